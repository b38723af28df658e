#!/bin/bash
for tx in 16; do for tz in 2 1; do
echo "== TX=$tx TZ=$tz"
BMV_CONV_SPLIT_TX=$tx BMV_CONV_SPLIT_TZ=$tz timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -k split_bf16 2>&1 | tail -1
BMV_CONV_SPLIT_TX=$tx BMV_CONV_SPLIT_TZ=$tz python scripts/bench_conv_split.py 2>&1 | grep -v amdgpu
done; done
