#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5u; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -q -x -k "sweep" 2>&1 | tail -3
python3 scripts/bench_sweep_quad.py --variants 12,17,18,0,19,20 --iters 40 > $O/plane_walk_warm.txt 2>&1
python3 scripts/bench_sweep_quad.py --variants 12,17,18,0,19,20 --iters 40 --evict-mb 32 > $O/plane_walk_evict32.txt 2>&1
grep -v "amdgpu.ids" $O/plane_walk_warm.txt; echo; grep -v "amdgpu.ids" $O/plane_walk_evict32.txt
python3 bench.py --no-cpu-baseline --steps 300 > $O/b.json 2> $O/b.err
python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('frame', round(d['value'],2), round(d['ms_per_step'],4), 'resident', round(d['value_extra']['resident_batch']['value'],2), {k: (round(v['avg_us'],1), round(v['frac'],3)) for k,v in d['roofline']['levels'].items()})"
