#!/bin/bash
mkdir -p gpurun_out/r3q
timeout 300 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_boost.py -x -q -m gpu 2>&1 | tail -3
show() { python -c "
import json,sys
d=json.loads(open('$1').read())
print('$2 value %.1f ms %.4f' % (d['value'], d['ms_per_step']), '| eager %.1f' % d['value_extra'].get('sync_bracketed_eager',{}).get('value',0), '| pipelined %.1f' % d['value_extra'].get('pipelined_replay',{}).get('value',0), '| cold %.1f' % d['value_extra'].get('value_cold',{}).get('value',0), '| step_ms', {k:round(v,3) for k,v in d['value_extra'].get('step_ms',{}).items() if k!='what'})
"; }
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3q/b1.json; show gpurun_out/r3q/b1.json default
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3q/b2.json; show gpurun_out/r3q/b2.json default
BMV_AUTOGRAPH=0 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3q/b3.json; show gpurun_out/r3q/b3.json autograph0
timeout 600 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 > gpurun_out/r3q/b4.json; show gpurun_out/r3q/b4.json noevents
