#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_mvs; mkdir -p $O; rm -f $O/ab.txt
python3 -m pytest tests -q -m gpu -k "mvs" 2>&1 | tail -6
for v in 0 1 0 1; do
  BMV_MVS_SPLIT=$v python3 bench.py --workload mvsnerf_ours_224x352_128planes_k4 --no-cpu-baseline --steps 4 --warmup 2 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); m=d.get('roofline_mfma',{}); print('mvs split $v', round(d['value'],4), round(d['ms_per_step'],2), 'renderer us', m.get('avg_us'))" | tee -a $O/ab.txt
done
