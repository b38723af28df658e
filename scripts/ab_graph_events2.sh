#!/bin/bash
for i in 1 2 3 4; do
  python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py every-4th | cut -c1-40
  python3 bench.py --no-cpu-baseline --cut-sweeps 2>/dev/null | python3 scripts/bench_line.py cut | cut -c1-40
done
