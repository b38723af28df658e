mkdir -p gpurun_out/r4l
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q > gpurun_out/r4l/pytest.txt 2>&1
tail -3 gpurun_out/r4l/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4l/bench.json 2> gpurun_out/r4l/bench.err
