#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5t; mkdir -p $O
for v in 0 1 0 1 0 1; do
  BMV_QUAD_S0=$v python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('quad s0 $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2))" | tee -a $O/ab.txt
done
