#!/bin/bash
set -u
O=gpurun_out/r3m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mvs.py -x -q > $O/mvs.log 2>&1; echo "mvs rc=$?"; tail -15 $O/mvs.log | cut -c1-250
timeout 300 python -m pytest tests/test_gpu_training.py -x -q -k graphed > $O/tr.log 2>&1; echo "graphed rc=$?"; tail -3 $O/tr.log
