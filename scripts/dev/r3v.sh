#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity.py -x -q -k "fpn or network_forward" 2>&1 | tail -2
BMV_FPN_SMOOTH_PERSIST=3 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q -k "fpn or whole_frame" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],4), round(d['value_extra']['step_ms']['median'],4))"; }
run base
BMV_FPN_SMOOTH_PERSIST=3 run persist3
BMV_FPN_SMOOTH_PERSIST=2 run persist2
BMV_FPN_SMOOTH_PERSIST=3 BMV_SIDE_PRIO=-1 run persist3+prio
BMV_FPN_SMOOTH_PERSIST=4 run persist4
run base
