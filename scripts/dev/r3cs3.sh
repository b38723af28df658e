#!/bin/bash
for s in auto 0 3 auto 0; do
BMV_CONV_SPLIT=$s python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s', round(d['value'],1), round(d['ms_per_step'],4), 'median', round(d['value_extra']['step_ms']['median'],4), 'leg(2-piece)', round(d['value_extra'].get('split_bf16_first_last_layers',{}).get('value',0),1))"
done
