#!/bin/bash
set -u
O=gpurun_out/r3_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
python bench.py > $O/default_bench.json 2> $O/default_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/default_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["frac"], {k:round(v["frac"],3) for k,v in d["roofline"]["levels"].items()}, "mfma", d["roofline_mfma"]["frac"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity_max_rel"]["max"], "split", d["value_extra"].get("split_bf16_first_last_layers",{}).get("value"), (d.get("parity_max_rel_split") or {}).get("max"))
print({k:(round(v["value"],1) if isinstance(v,dict) and "value" in v else None) for k,v in d["value_extra"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
