#!/bin/bash
# FeatureNet's first block as one launch (3-channel first layer in the second layer's tile producer) vs two launches
for i in 1 2 3; do
  BMV_CONV0_FUSE=0 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py two-launches | cut -c1-90
  BMV_CONV0_FUSE=1 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py fused-r4 | cut -c1-90
  BMV_CONV0_FUSE=1 BMV_CONV0_R=8 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py fused-r8 | cut -c1-90
done
