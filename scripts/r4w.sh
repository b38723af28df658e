mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_framegraph.py -x -q > gpurun_out/r4w/ta.txt 2>&1; tail -3 gpurun_out/r4w/ta.txt | cut -c1-160
timeout 600 python scripts/probe_autograph_cost.py > gpurun_out/r4w/probe.txt 2>&1
