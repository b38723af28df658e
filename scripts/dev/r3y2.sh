#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_framegraph.py -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload enerf_ours_480x736_6src_k4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['value'],2), round(d['ms_per_step'],4))"
BMV_FRAME_SETUP=0 python bench.py --workload enerf_ours_480x736_6src_k4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 no frame_setup', round(d['value'],2), round(d['ms_per_step'],4))"
done
