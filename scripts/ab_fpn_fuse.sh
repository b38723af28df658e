#!/bin/bash
# FeatureNet's last top-down step fused into smooth0 (default) vs the two launches, 8- and 4-row tiles
for i in 1 2 3; do
  BMV_FPN_FUSE=0 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py two-launches | cut -c1-36
  BMV_FPN_FUSE=1 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py fused-r8 | cut -c1-36
  BMV_FPN_FUSE=1 BMV_FPN_SMOOTH_R=4 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py fused-r4 | cut -c1-36
done
