#!/usr/bin/env python3
"""One steady-state frame of a rocprofv3 kernel trace, launch by launch: start offset, duration, gap to the previous
end, workgroups, threads, LDS -- the view that shows which short launches leave the chip idle.
    python scripts/frame_timeline.py <kernel_trace.csv> [--marker render_pc_kernel,render_rays_kernel] [--frame -2]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="render_pc_kernel,render_rays_kernel",
                help="comma-separated kernel-name fragments; the last launch of a frame matches one of them")
ap.add_argument("--frame", type=int, default=-2)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in a.marker.split(","))]
lo, hi = marks[a.frame - 1] + 1, marks[a.frame] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
print(f"{'t us':>8} {'dur':>7} {'gap':>6} {'wgs':>6} {'thr':>4} {'lds':>6} {'vgpr':>4}  kernel")
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    vg = 2 * (int(r.get("Arch_VGPR_Count", r.get("VGPR_Count", 0)) or 0))
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f} {grid // wg:6d} {wg:4d} "
          f"{int(r.get('LDS_Block_Size', 0) or 0):6d} {vg:4d}  {r['Kernel_Name'][:80]}")
    prev_end = max(prev_end, e)
print(f"frame span {(prev_end - t0) / 1e3:.1f} us")
