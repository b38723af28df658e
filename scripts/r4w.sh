mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_boost.py -x -q > gpurun_out/r4w/ta.txt 2>&1; tail -5 gpurun_out/r4w/ta.txt | cut -c1-180
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4w/bench_c2.json 2> gpurun_out/r4w/bench_c2.err
timeout 600 python bench.py --no-cpu-baseline --workload enerf_ours_480x736_6src_k4 > gpurun_out/r4w/bench_c3.json 2> gpurun_out/r4w/bench_c3.err
