#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 300 python -m pytest tests/test_gpu_mvs.py tests/test_gpu_configs34.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -12
for aux in 2 0; do
BMV_MVS_SWEEP_AUX=$aux timeout 600 python bench.py --workload mvsnerf_ours_224x352_128planes_k4 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('aux $aux value', round(d['value'],3), 'sweep', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('avg_us','frac')})"
done
