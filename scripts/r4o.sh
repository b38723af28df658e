mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_boost.py tests/test_gpu_configs34.py -x -q > gpurun_out/r4o/pytest.txt 2>&1; tail -4 gpurun_out/r4o/pytest.txt
python scripts/bench_blend.py > gpurun_out/r4o/blend.txt 2>&1
