#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r3s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sb
rocprofv3 --kernel-trace --stats -d /tmp/prof_sb --output-format csv -- python3 $R/scripts/bench_sweep_bwd.py > $O/prof_run.txt 2>&1
T=$(ls /tmp/prof_sb/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$T")) if "sweep_bwd" in r["Kernel_Name"]]
for r in rows:
    print(r["Kernel_Name"][:60], r["Grid_Size_X"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
