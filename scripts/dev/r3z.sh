#!/bin/bash
set -u
O=gpurun_out/r3z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_conv.py tests/test_gpu_boost.py tests/test_gpu_mvs.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 900 python scripts/profile_train_ops.py enerf_ours_ft_480x736_6src_k4 50 > $O/c5_ops.txt 2>&1; echo "ops rc=$?"
timeout 900 python scripts/profile_train_ops.py enerf_ours_ft_480x736_6src_k4 30 --stacks > $O/c5_ops_stacks.txt 2>&1; echo "stacks rc=$?"
PFX=r3 timeout 1200 bash scripts/collect_profiles.sh $O c5 ft
