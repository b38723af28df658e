#!/bin/bash
# MFMA-pipe and LDS counters per kernel over eager frames of the headline workload (one --pmc pass).
#   bash scripts/pmc_frame.sh <outdir>          (WORKLOAD=<bench workload> STEPS=<n> for another frame, e.g. BASELINE configs[3])
OUT=${1:-gpurun_out/pmc_frame}; R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_frame
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_frame --output-format csv -- python3 $R/bench.py ${WORKLOAD:+--workload $WORKLOAD} --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --graph 0 --spinup-steps 0 > /tmp/pmc_frame.out 2>&1
F=$(ls /tmp/pmc_frame/*/*counter_collection.csv | head -1)
python3 - "$F" > $R/$OUT/frame_mfma_lds_counters.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:78]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':78s} {'launches':>8s} {'MFMA busy':>9s} {'LDS conflict / active':>22s}")
out = []
for k, c in agg.items():
    n = len(c.get("SQ_BUSY_CYCLES", []))
    busy = sum(c.get("SQ_BUSY_CYCLES", [0])) / max(n, 1)
    mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(n, 1)
    conf = sum(c.get("SQ_LDS_BANK_CONFLICT", [0])) / max(n, 1)
    act = sum(c.get("SQ_LDS_IDX_ACTIVE", [0])) / max(n, 1)
    # MFMA_BUSY counts per SIMD (1024), BUSY per SQ (32 x 8 XCD = 256 ... normalised as in round 1: /1024 vs /32)
    frac = (mf / 1024.0) / (busy / 32.0) if busy else 0.0
    out.append((busy, k, n, frac, conf / act if act else 0.0))
for busy, k, n, frac, cf in sorted(out, reverse=True)[:40]:
    print(f"{k:78s} {n:8d} {frac:9.2f} {cf:22.2f}")
PY
cp $F $R/$OUT/frame_mfma_lds_counters.csv
cat $R/$OUT/frame_mfma_lds_counters.txt | cut -c1-125
