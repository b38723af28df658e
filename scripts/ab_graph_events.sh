#!/bin/bash
# event brackets of the roofline kernels under graph replay: on every 4th step (default), on every step, as graph cuts
# around the sweeps (round 1), none at all
for i in 1 2; do
  python3 bench.py --steps 32 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py every-4th
  python3 bench.py --steps 32 --warmup 5 --no-cpu-baseline --event-every 1 2>/dev/null | python3 scripts/bench_line.py every-step
  python3 bench.py --steps 32 --warmup 5 --no-cpu-baseline --cut-sweeps 2>/dev/null | python3 scripts/bench_line.py cut
  python3 bench.py --steps 32 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python3 scripts/bench_line.py none
done
python3 bench.py --workload enerf_ours_480x736_6src_k4 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 | python3 scripts/bench_line.py boost
python3 bench.py --workload enerf_ours_480x736_6src_k4 --steps 12 --warmup 3 --no-cpu-baseline --cut-sweeps 2>&1 | tail -3 | python3 scripts/bench_line.py boost-cut
