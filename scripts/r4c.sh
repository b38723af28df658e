mkdir -p gpurun_out/r4c
./scripts/ubench/sweep_sol2 8 > gpurun_out/r4c/sweep_sol2.txt 2>&1
