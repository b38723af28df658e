#!/bin/bash
for aux in 2 258 256 0; do
BMV_MVS_SWEEP_AUX=$aux timeout 300 python -m pytest tests/test_gpu_mvs.py -x -q -k "proj_resize_sweep" 2>&1 | tail -1
BMV_MVS_SWEEP_AUX=$aux python bench.py --workload mvsnerf_ours_224x352_128planes_k4 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('aux $aux', round(d['value'],3), {n:round(v['avg_us'],1) for n,v in k.items() if 'sweep' in n})"
done
