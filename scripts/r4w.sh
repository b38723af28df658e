mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_framegraph.py -x -q > gpurun_out/r4w/ta.txt 2>&1; tail -15 gpurun_out/r4w/ta.txt | cut -c1-180
timeout 1500 python -m pytest tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4w/tb.txt 2>&1; tail -3 gpurun_out/r4w/tb.txt | cut -c1-160
timeout 600 python scripts/probe_autograph_cost.py > gpurun_out/r4w/probe.txt 2>&1
BMV_AUTOGRAPH_RING=0 timeout 600 python scripts/probe_autograph_cost.py > gpurun_out/r4w/probe_noring.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4w/bench_c2.json 2> gpurun_out/r4w/bench_c2.err
