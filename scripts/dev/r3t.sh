#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_boost.py tests/test_gpu_configs34.py tests/test_gpu_framegraph.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -12
for bt in 1 0; do
BMV_BOOST_BATCHED=$bt timeout 600 python bench.py --workload enerf_ours_480x736_6src_k4 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batched $bt value', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'eager', round(d['value_extra']['sync_bracketed_eager']['value'],1))"
done
