#!/bin/bash
for cfg in "32 2 1" "16 2 1" "32 1 1"; do set -- $cfg
echo "== TX=$1 TZ=$2 RW=$3"
BMV_CONV_SPLIT_TX=$1 BMV_CONV_SPLIT_TZ=$2 BMV_CONV_SPLIT_RW=$3 timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -k split_bf16 2>&1 | tail -1
BMV_CONV_SPLIT_TX=$1 BMV_CONV_SPLIT_TZ=$2 BMV_CONV_SPLIT_RW=$3 python scripts/bench_conv_split.py 2>&1 | grep -v amdgpu
done
