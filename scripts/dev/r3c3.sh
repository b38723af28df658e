#!/bin/bash
run() { python bench.py --workload enerf_ours_480x736_6src_k4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4))"; }
run base
BMV_RENDER_PC_GRID=128 run grid128
BMV_RENDER_PC_GRID=64 run grid64
BMV_RENDER_PC=0 run fused_renderer
BMV_CONV_SPLIT=2 run split2
run base
