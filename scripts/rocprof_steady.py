#!/usr/bin/env python3
"""Per-kernel statistics of the STEADY-STATE steps of a bench.py run from a rocprofv3 kernel trace.

MIOpen's solver search (torch.backends.cudnn.benchmark) runs reference convolutions during the first
steps; `rocprofv3 --stats` sums them in.  This script keeps only the dispatches after the
(n_steps+1)-th last occurrence of the marker kernel (one per step) and prints / writes the table.

    python scripts/rocprof_steady.py <..._kernel_trace.csv> [--marker render_rays_kernel] [--steps 8] [--out x.csv]
"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="render_rays_kernel")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--out", default=None)
ap.add_argument("--markers-per-step", type=int, default=1, help="dispatches of the marker kernel in one step")
a = ap.parse_args()

rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]][a.markers_per_step - 1::a.markers_per_step]
if len(marks) < a.steps + 1:
    raise SystemExit(f"only {len(marks)} marker dispatches")
lo, hi = marks[-(a.steps + 1)] + 1, marks[-1] + 1          # a.steps whole steps, ending on a marker
agg = collections.defaultdict(list)
for r in rows[lo:hi]:
    agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
span = int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
table = sorted(((sum(v), k, len(v)) for k, v in agg.items()), reverse=True)
print(f"steady steps: {a.steps}   kernel time/step {tot / a.steps / 1e3:.1f} us   wall/step {span / a.steps / 1e3:.1f} us")
out = [("Name", "CallsPerStep", "AvgUs", "UsPerStep", "Percent")]
for t, k, n in table:
    out.append((k, n / a.steps, t / n / 1e3, t / a.steps / 1e3, 100.0 * t / tot))
for row in out[1:25]:
    print(f"{row[0][:90]:90s} {row[1]:6.1f} {row[2]:9.1f} {row[3]:9.1f} {row[4]:5.1f}%")
if a.out:
    with open(a.out, "w", newline="") as f:
        csv.writer(f).writerows(out)
