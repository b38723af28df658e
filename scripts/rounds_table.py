"""Per kernel of a rocprofv3 --kernel-trace csv: workgroups, resident workgroups per CU (from VGPRs / LDS / the
16-waves-per-SIMD... limit of 8 here), and how many 'rounds' of the chip the grid is -- a grid that is 1.1 rounds
leaves the machine mostly idle for almost half its run."""
import csv
import sys
from collections import defaultdict


def main(path, skip_first=0.3):
    rows = list(csv.DictReader(open(path)))
    rows = rows[int(len(rows) * skip_first):]          # drop warm-up launches
    agg = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        # rocprofv3 (ROCm 7.2) reports half the allocated VGPRs here (96 for a 191-register kernel)
        vg = 2 * (int(r.get("Arch_VGPR_Count", r.get("VGPR_Count", 0)) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0))
        lds = int(r.get("LDS_Block_Size", 0) or 0)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        agg[(name, grid, wg, vg, lds)].append(dur)
    out = []
    for (name, grid, wg, vg, lds), d in agg.items():
        nwg = grid // wg
        waves = max(wg // 64, 1)
        by_v = (512 // max(-(-vg // 8) * 8, 8)) if vg else 8
        by_v = min(by_v, 8)
        per_cu = min(by_v * 4 // waves, (160 * 1024) // lds if lds else 99, 32)
        per_cu = max(per_cu, 1)
        out.append((sum(d) / len(d), name[:70], nwg, wg, vg, lds, per_cu, nwg / (256.0 * per_cu), len(d)))
    out.sort(reverse=True)
    print(f"{'avg us':>8} {'wgs':>7} {'thr':>4} {'vgpr':>4} {'lds':>6} {'wg/cu':>5} {'rounds':>6}  kernel")
    for avg, name, nwg, wg, vg, lds, per_cu, rounds, n in out[:40]:
        print(f"{avg:8.1f} {nwg:7d} {wg:4d} {vg:4d} {lds:6d} {per_cu:5d} {rounds:6.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
