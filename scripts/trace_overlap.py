#!/usr/bin/env python3
"""Steady-state frame timeline from a rocprofv3 kernel trace: wall per frame, sum of kernel durations,
time with >= 2 kernels in flight (stream / graph-branch overlap), idle gaps.
    python scripts/trace_overlap.py <kernel_trace.csv> [--marker render_rays_kernel] [--steps 8]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="render_rays_kernel")
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
lo, hi = marks[-(a.steps + 1)] + 1, marks[-1] + 1
ev = []
for r in rows[lo:hi]:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = over = idle = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    dt = t - last
    if depth == 0:
        idle += dt
    else:
        busy += dt
        if depth >= 2:
            over += dt
    depth += d
    last = t
span = ev[-1][0] - ev[0][0]
ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo:hi])
n = a.steps
print(f"per frame: wall {span / n / 1e3:.1f} us, kernel sum {ksum / n / 1e3:.1f} us, busy {busy / n / 1e3:.1f} us, "
      f">=2 kernels in flight {over / n / 1e3:.1f} us, idle {idle / n / 1e3:.1f} us, launches {(hi - lo) / n:.1f}")
