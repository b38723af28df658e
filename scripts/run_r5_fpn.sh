#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_fpn; mkdir -p $O; rm -f $O/ab.txt
run() { # label, env...
  env "${@:2}" python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2))" | tee -a $O/ab.txt
}
run "default" A=1
run "fpn_smooth R=4" BMV_FPN_SMOOTH_R=4
run "fpn_smooth persist 2" BMV_FPN_SMOOTH_PERSIST=2
run "fpn_smooth persist 3" BMV_FPN_SMOOTH_PERSIST=3
run "default" A=1
run "topdown split 2" BMV_FPN_TOPDOWN_SPLIT=2
run "topdown split 4" BMV_FPN_TOPDOWN_SPLIT=4
run "conv0 R=8" BMV_CONV0_R=8
run "default" A=1
