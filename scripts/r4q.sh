mkdir -p gpurun_out/r4q
timeout 1200 python -m pytest tests/test_gpu_training.py -x -q -k "source_views or finetune_gradients" > gpurun_out/r4q/pytest.txt 2>&1; tail -25 gpurun_out/r4q/pytest.txt
