#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r3n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 32768 131072; do
rm -rf /tmp/prof_mlp
rocprofv3 --kernel-trace --stats -d /tmp/prof_mlp --output-format csv -- python3 $R/scripts/bench_mvs_mlp_train.py $n > $O/bench_$n.txt 2>&1
S=$(ls /tmp/prof_mlp/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
print("== $n points")
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
done
