mkdir -p gpurun_out/r4n
timeout 900 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_boost.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4n/pytest.txt 2>&1; tail -5 gpurun_out/r4n/pytest.txt
python scripts/probe_autograph_cost.py > gpurun_out/r4n/probe.txt 2>&1
BMV_AUTOGRAPH_IO=0 python scripts/probe_autograph_cost.py > gpurun_out/r4n/probe_noio.txt 2>&1
