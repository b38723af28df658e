mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -x -k "quad or sweep" 2>&1 | tail -3 > gpurun_out/r4w/pytest.txt
timeout 600 python scripts/bench_sweep_quad.py --variants 0,12 --iters 60 > gpurun_out/r4w/quad.txt 2>&1
timeout 600 python scripts/bench_sweep_quad.py --variants 0,12 --iters 60 --evict-mb 32 > gpurun_out/r4w/quad_evict.txt 2>&1
