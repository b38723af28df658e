#!/bin/bash
for tz in 2 1; do
echo "== TZ=$tz"
BMV_CONV_SPLIT_TZ=$tz timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -s -k split_bf16 2>&1 | grep "conv_split\|passed\|failed" | cut -c1-150
BMV_CONV_SPLIT_TZ=$tz python scripts/bench_conv_split.py 2>&1 | grep -v amdgpu
done
