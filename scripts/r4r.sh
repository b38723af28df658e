mkdir -p gpurun_out/r4r
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "config3 or config4 or whole_frame" > gpurun_out/r4r/pytest.txt 2>&1; grep -n "^\[config\|passed\|failed\|Error\|error" gpurun_out/r4r/pytest.txt | head -30; tail -5 gpurun_out/r4r/pytest.txt
