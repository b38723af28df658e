#!/bin/bash
O=gpurun_out/r3cs; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_auto.log 2>&1; echo "suite default(auto) rc=$?"; tail -4 $O/suite_auto.log | cut -c1-200
for s in auto 0 auto 0; do
BMV_CONV_SPLIT=$s python bench.py --cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s', round(d['value'],1), round(d['ms_per_step'],4), 'parity', d.get('parity_max_rel',{}).get('max'), 'leg', round(d['value_extra'].get('split_bf16_first_last_layers',{}).get('value',0),1))"
done
