#!/bin/bash
set -u
O=gpurun_out/r3z; mkdir -p $O
timeout 300 python scripts/dev/r3_gradnoise.py > $O/gradnoise.txt 2>&1; echo "noise rc=$?"; cat $O/gradnoise.txt | tail -24
timeout 900 python scripts/profile_train_ops.py --workload enerf_ours_ft_480x736_6src_k4 --rows 45 > $O/c5_ops.txt 2>&1; echo "ops rc=$?"
