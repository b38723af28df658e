#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_mvs.py -x -q 2>&1 | tail -2
python scripts/bench_mvs_mlp_train.py 131072 2>&1 | grep forward; python scripts/bench_mvs_mlp_train.py 32768 2>&1 | grep forward
