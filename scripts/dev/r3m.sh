#!/bin/bash
for z in 1 0 1; do echo "== ZP=$z"; BMV_SWEEP_ZP=$z timeout 600 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "test_boost_enerf_finetune_gradients" 2>&1 | grep -E "passed|failed|Error|assert|outlier|worst" | head -12; done
