mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_boost.py tests/test_gpu_configs34.py -x -q > gpurun_out/r4w/ta.txt 2>&1; tail -5 gpurun_out/r4w/ta.txt | cut -c1-180
