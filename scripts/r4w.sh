mkdir -p gpurun_out/r4w
timeout 120 ./scripts/ubench/grid_barrier > gpurun_out/r4w/grid_barrier.txt 2>&1
