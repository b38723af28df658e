#!/bin/bash
mkdir -p gpurun_out/r3l
for z in 1 0 1 0; do
  BMV_SWEEP_ZP=$z timeout 300 python bench.py --steps 100 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r3l/bench_zp$z.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r3l/bench_zp$z.json"))
print("zp=$z value", d["value"], "ms", d["ms_per_step"], {k:(v.get("us"),v.get("frac")) for k,v in d["roofline"].get("levels",{}).items()})
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r3l/pytest.log
