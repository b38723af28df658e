#!/bin/bash
mkdir -p gpurun_out/r3p
timeout 600 python bench.py 2>gpurun_out/r3p/bench.err | tail -1 > gpurun_out/r3p/bench.json
tail -3 gpurun_out/r3p/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3p/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], d["config"]["launch"])
for k,v in d["value_extra"].items(): print(" ", k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="what"})
print("roofline", d["roofline"]["frac"], {k:(round(v["avg_us"],2), round(v["frac"],3)) for k,v in d["roofline"]["levels"].items()}, d["roofline"].get("traffic_source"))
print("mfma", {k:v for k,v in d["roofline_mfma"].items() if k!="kernel"})
print("parity", d.get("parity_max_rel"))
print("cpu", d.get("cpu_baseline"))
PY
timeout 300 python bench.py --workload enerf_256x320_3src_32planes --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 value', d['value'], d['ms_per_step'], d['value_extra'].get('sync_bracketed_eager',{}).get('value'))"
timeout 300 python bench.py --workload enerf_ours_480x736_6src_k4 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 value', d['value'], d['ms_per_step'], d['config']['launch'], d['value_extra'].get('sync_bracketed_eager',{}).get('value'))"
