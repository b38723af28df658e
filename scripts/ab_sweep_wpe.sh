#!/bin/bash
# A/B: windowed sweep compiled for 4 (default) vs 5 waves per SIMD.  10240 waves of work on 4096 slots is 2.5 rounds;
# on 5120 slots exactly 2.  Run on the MI355X box from the repo root.
R=$(pwd)
for wpe in 4 5; do
  rm -f $R/boostmvsnerfs_amd/csrc/sweep_win.o
  BMV_WIN_DEFS="-DBMV_WIN_WPE=$wpe" python -m boostmvsnerfs_amd.build > /tmp/build_wpe$wpe.log 2>&1
  echo "== WPE $wpe"
  python scripts/tune_sweep_win.py --iters 100 --variants 0,14,9,12 2>&1 | grep -E "level|variant"
  BMV_SWEEP_WIN_CAP=240 python scripts/tune_sweep_win.py --iters 100 --variants 12 2>&1 | grep -E "variant" | sed 's/^/  cap240 /'
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/wpe.json
  python - <<'PY'
import json
d = json.loads(open('/tmp/wpe.json').read())
print("  frame", round(d["value"], 1), {k: (round(v["avg_us"], 2), round(v["frac"], 3)) for k, v in d["roofline"]["levels"].items()})
PY
done
rm -f $R/boostmvsnerfs_amd/csrc/sweep_win.o
python -m boostmvsnerfs_amd.build > /dev/null 2>&1
