#!/bin/bash
# FeatureNet's conv2.1 + toplayer as one launch vs two
for i in 1 2 3 4; do
  for m in 0 1; do
    BMV_TOP_FUSE=$m python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py top_fuse=$m | cut -c1-90
  done
done
