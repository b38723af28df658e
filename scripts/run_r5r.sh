#!/bin/bash
# quad-record cost volumes between the sweep and the regularisers' first layer: tests, then the frame A/B
R=$(pwd); O=$R/gpurun_out/r5r; mkdir -p $O
python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_parity.py tests/test_gpu_framegraph.py tests/test_gpu_configs34.py -q -x 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
for v in 0 1 0 1; do
  BMV_QUAD_VOLUME=$v python3 bench.py --no-cpu-baseline --steps 400 > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('quad volume $v', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), {k: (round(v['avg_us'],1), round(v['frac'],3)) for k,v in d['roofline']['levels'].items()})" | tee -a $O/ab.txt
done
for v in 0 1; do
  BMV_QUAD_VOLUME=$v python3 bench.py --no-cpu-baseline --workload enerf_ours_480x736_6src_k4 --steps 60 > $O/b3.json 2> $O/b3.err
  python3 -c "
import json; d=json.loads(open('$O/b3.json').read().strip().splitlines()[-1]); print('config 3 quad volume $v', round(d['value'],2), round(d['ms_per_step'],4))" | tee -a $O/ab.txt
done
