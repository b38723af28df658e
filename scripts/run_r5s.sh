#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5s; mkdir -p $O
python3 scripts/bench_conv_c4.py > $O/conv_c4_layers.txt 2>&1; cat $O/conv_c4_layers.txt
python3 -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('frame', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2))"
