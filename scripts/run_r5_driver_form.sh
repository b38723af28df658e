#!/bin/bash
# the driver's command forms
R=$(pwd); O=$R/gpurun_out/r5_driver; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 10 --warmup 3 > $O/a.json 2> $O/a.err ) 2>&1 | tail -3
python3 -c "
import json; d=json.loads(open('$O/a.json').read().strip().splitlines()[-1]); print('steps 10:', round(d['value'],1), d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup'], 'fp32 extra', d['value_extra'].get('renderer_fp32_mfma',{}).get('value'))"
( time python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > $O/b.json 2> $O/b.err ) 2>&1 | tail -3
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('torchrun form:', round(d['value'],1), d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup'])"
tail -3 $O/b.err
