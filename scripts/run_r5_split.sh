#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O; rm -f $O/ab6.txt
python3 tests/tools/mlp_split_accuracy.py 2>&1 | grep -v amdgpu.ids | tee $O/mlp_split_accuracy_rn.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_cabi.py -q -x 2>&1 | tail -3
for v in 0 1 0 1; do
  BMV_RENDER_SPLIT=$v python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); m=d.get('roofline_mfma',{}); x=d['value_extra']
print('render split $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(x['resident_batch']['value'],2), 'pipelined', round(x.get('pipelined_replay',{}).get('value',0),1), 'renderer us', round(m.get('avg_us'),1))" | tee -a $O/ab6.txt
done
