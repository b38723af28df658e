#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O
python3 tests/tools/mlp_split_accuracy.py 2>&1 | grep -v amdgpu.ids | tee $O/mlp_split_accuracy.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_cabi.py -q -x 2>&1 | tail -3
python3 bench.py --steps 400 > $O/bench_with_split_extra.json 2> $O/b.err
python3 -c "
import json; d=json.loads(open('$O/bench_with_split_extra.json').read().strip().splitlines()[-1]); print('value', round(d['value'],2), 'parity', d['parity_max_rel']['max']); print(json.dumps(d['value_extra'].get('render_split_bf16x3'), indent=1))"
