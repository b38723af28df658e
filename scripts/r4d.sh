mkdir -p gpurun_out/r4d
python scripts/bench_sweep_quad.py > gpurun_out/r4d/quad.txt 2>&1
python scripts/bench_sweep_quad.py --cold > gpurun_out/r4d/quad_cold.txt 2>&1
