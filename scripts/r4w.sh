mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/dbg_defer.py -x -q -s > gpurun_out/r4w/ta.txt 2>&1; grep -a "passed\|Abort\|failed" gpurun_out/r4w/ta.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_framegraph.py -x -q > gpurun_out/r4w/tc.txt 2>&1; tail -3 gpurun_out/r4w/tc.txt | cut -c1-160
timeout 1500 python -m pytest tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r4w/tb.txt 2>&1; tail -3 gpurun_out/r4w/tb.txt | cut -c1-160
timeout 600 python scripts/probe_autograph_cost.py > gpurun_out/r4w/probe.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4w/bench_c2.json 2> gpurun_out/r4w/bench_c2.err
timeout 600 python bench.py --workload enerf_ours_480x736_6src_k4 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4w/bench_c3.json 2> gpurun_out/r4w/bench_c3.err
