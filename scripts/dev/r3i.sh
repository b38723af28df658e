#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_ft
rocprofv3 --kernel-trace -d /tmp/tl_ft --output-format csv -- python3 $R/bench.py --workload enerf_ft_512x640_3src --steps 4 --warmup 6 --no-cpu-baseline > /tmp/tl_ft.out 2> /tmp/tl_ft.err
T=$(ls /tmp/tl_ft/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv
rows=sorted(csv.DictReader(open("$T")), key=lambda r:int(r["Start_Timestamp"]))
names=("img_feat_bwd","vox_feat_bwd","sweep_bwd","nerf_mlp_bwd","nerf_wgrad_kernel","img_feat_kernel","build_rays_bwd","bn_bwd_reduce")
last={}
for r in rows[-1200:]:
    for n in names:
        if n in r["Kernel_Name"]:
            last.setdefault(n,[]).append(((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]))
for n,v in last.items():
    print(n, [(round(a),g) for a,g in v[-6:]])
PY
