#!/bin/bash
# brackets of the roofline kernels under graph replay: sweeps between graphs with events bound to their dispatch
# (default), sweeps as event-record nodes inside one graph, no brackets at all
for i in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py bound
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-graph-sweeps 2>/dev/null | python3 scripts/bench_line.py in-graph
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python3 scripts/bench_line.py none
done
