set -x
mkdir -p gpurun_out/r4a
./scripts/ubench/sweep_sol 8 > gpurun_out/r4a/sweep_sol.txt 2>&1
./scripts/ubench/sweep_sol 1 > gpurun_out/r4a/sweep_sol_1out.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_training.py tests/test_gpu_mvs.py -x -q > gpurun_out/r4a/pytest_a.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -5 gpurun_out/r4a/pytest_a.txt
