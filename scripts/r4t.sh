mkdir -p gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lost_wakeup or tuning or quad_channel" > gpurun_out/r4t/pytest.txt 2>&1; tail -5 gpurun_out/r4t/pytest.txt
bash scripts/collect_profiles.sh gpurun_out/r4_profiles_a c3 c4 > gpurun_out/r4t/collect.log 2>&1
