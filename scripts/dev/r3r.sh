#!/bin/bash
mkdir -p gpurun_out/r3r
timeout 600 python -m pytest tests/test_gpu_mvs.py tests/test_gpu_fullsize.py tests/test_gpu_configs34.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8
for aux in 2 0; do
BMV_MVS_SWEEP_AUX=$aux timeout 600 python bench.py --workload mvsnerf_ours_224x352_128planes_k4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3r/cfg4_aux$aux.json
python - <<PY
import json
d=json.load(open("gpurun_out/r3r/cfg4_aux$aux.json"))
print("aux=$aux value", round(d["value"],3), "ms", round(d["ms_per_step"],2), "sweep", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("avg_us","frac","achieved","launches")})
print("   mfma", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline_mfma"].items() if k in ("avg_us","frac")})
PY
done
timeout 600 python bench.py --workload mvsnerf_224x352_32planes --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mvsnerf32 value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
