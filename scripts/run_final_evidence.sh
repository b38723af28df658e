#!/bin/bash
# Per-round evidence from the final tree (PFX=r6 by default): per-workload bench lines + rocprofv3 steady summaries
# (scripts/collect_profiles.sh), the S = 2 / 4 frames, the default `python bench.py` line.  Copy what is to be judged from
# gpurun_out/${PFX}_final/profiles into profiles/${PFX}/.      bash scripts/run_final_evidence.sh [workloads...]
PFX=${PFX:-r6}
R=$(pwd); O=$R/gpurun_out/${PFX}_final; mkdir -p $O
WL=${*:-"c2 c1 c3 c4 c5 ft"}
PFX=$PFX bash scripts/collect_profiles.sh gpurun_out/${PFX}_final/profiles $WL > $O/collect.txt 2>&1
for w in enerf_512x640_2src_64planes enerf_512x640_4src_64planes; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 200 > $O/profiles/${PFX}_${w}_bench.json 2> $O/${w}.err
done
python3 bench.py > $O/profiles/${PFX}_default_bench_line.json 2> $O/default.err
tail -20 $O/collect.txt
