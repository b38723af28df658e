#!/bin/bash
# the whole GPU suite + the entry point's smoke + the default bench line, timed.   bash scripts/run_full_tests.sh [tag]
PFX=${PFX:-r6}
R=$(pwd); O=$R/gpurun_out/${PFX}_tests; mkdir -p $O
N=${1:-1}
python3 -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu_$N.txt
cat $O/pytest_gpu_$N.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5 | tee $O/smoke_$N.txt
( time python3 bench.py > $O/bench_default_$N.json 2> $O/bench_default_$N.err ) 2>&1 | tail -4 | tee $O/bench_time_$N.txt
python3 -c "
import json; d=json.loads(open('$O/bench_default_$N.json').read().strip().splitlines()[-1]); print('default bench', round(d['value'],2), d['unit'], round(d['ms_per_step'],4))"
