#!/bin/bash
for r in 0 4 0 4; do
BMV_CONV_PAIR_ROWS=$r timeout 600 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pair rows $r value %.1f ms %.4f median %.4f' % (d['value'], d['ms_per_step'], d['value_extra']['step_ms']['median']))"
done
