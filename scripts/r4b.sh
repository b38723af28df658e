set -x
mkdir -p gpurun_out/r4b
./scripts/ubench/sweep_sol 8 > gpurun_out/r4b/sweep_sol.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_framegraph.py -x -q > gpurun_out/r4b/pytest_a.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err
tail -3 gpurun_out/r4b/pytest_a.txt
