#!/bin/bash
# final-tree evidence (after the renderer-split experiment went in): headline bench line (with value_extra.render_split_bf16x3
# and its parity), the K-volume and small-frame workloads with the experiment switched on by hand
R=$(pwd); O=$R/gpurun_out/r5_final2; mkdir -p $O; rm -f $O/split_other_workloads.txt
python3 bench.py > $O/r5_default_bench_line_final.json 2> $O/default.err
python3 -c "
import json; d=json.loads(open('$O/r5_default_bench_line_final.json').read().strip().splitlines()[-1]); x=d['value_extra']; print('default line: value', round(d['value'],2), d['ms_per_step'], 'parity', d['parity_max_rel']['max'], 'split extra', x.get('render_split_bf16x3'))"
for w in enerf_ours_480x736_6src_k4 enerf_256x320_3src_32planes; do
  for v in 0 1 0 1; do
    BMV_RENDER_SPLIT=$v python3 bench.py --workload $w --no-cpu-baseline --steps 100 > $O/b.json 2> $O/b.err
    python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$w render split $v', round(d['value'],2), round(d['ms_per_step'],4))" | tee -a $O/split_other_workloads.txt
  done
done
