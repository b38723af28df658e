#!/bin/bash
O=gpurun_out/r3cs; mkdir -p $O
BMV_CONV_SPLIT=1 timeout 1500 python -m pytest tests -m gpu -q > $O/suite_split.log 2>&1; echo "suite with BMV_CONV_SPLIT=1 rc=$?"; tail -6 $O/suite_split.log | cut -c1-200
for s in 0 1 0 1; do
BMV_CONV_SPLIT=$s python bench.py --cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s', round(d['value'],1), round(d['ms_per_step'],4), 'parity', d.get('parity_max_rel',{}).get('max'), {k:round(v,7) for k,v in d.get('parity_max_rel',{}).get('per_output',{}).items()})"
done
