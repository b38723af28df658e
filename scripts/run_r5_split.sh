#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O; rm -f $O/ab5.txt
for v in 0 1 0 1 0 1; do
  BMV_RENDER_SPLIT=$v python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); m=d.get('roofline_mfma',{}); x=d['value_extra']
print('render split $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(x['resident_batch']['value'],2), 'pipelined', round(x.get('pipelined_replay',{}).get('value',0),1), 'renderer us', round(m.get('avg_us'),1), 'steps', x.get('step_ms'))" | tee -a $O/ab5.txt
done
