mkdir -p gpurun_out/r4u
timeout 900 python -m pytest tests/dbg_cam_pre.py -q -s 2>&1 | tail -12 > gpurun_out/r4u/dbg.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_fullsize.py -q -k "not whole_frame" > gpurun_out/r4u/pytest.txt 2>&1; tail -5 gpurun_out/r4u/pytest.txt
timeout 600 python bench.py --workload enerf_ours_480x736_6src_k4 --steps 100 --warmup 5 > gpurun_out/r4u/bench_c3.json 2> gpurun_out/r4u/bench_c3.err
