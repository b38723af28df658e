mkdir -p gpurun_out/r4i
python scripts/bench_sweep_quad.py > gpurun_out/r4i/quad.txt 2>&1
