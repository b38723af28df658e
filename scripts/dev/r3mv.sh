#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_mvs.py tests/test_gpu_configs34.py -x -q 2>&1 | tail -2
python scripts/bench_mvs_sweep.py 2>&1 | grep mvs_sweep
python bench.py --workload mvsnerf_ours_224x352_128planes_k4 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('c4', round(d['value'],3), {n:round(v['avg_us'],1) for n,v in k.items() if 'sweep' in n}, d['roofline']['frac'])"
