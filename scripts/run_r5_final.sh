#!/bin/bash
# round-5 evidence from the final tree: per-workload bench lines + rocprofv3 steady summaries, the S = 2 / 4 frames
R=$(pwd); O=$R/gpurun_out/r5_final; mkdir -p $O
PFX=r5 bash scripts/collect_profiles.sh gpurun_out/r5_final/profiles c2 c1 c3 c4 c5 ft > $O/collect.txt 2>&1
for w in enerf_512x640_2src_64planes enerf_512x640_4src_64planes; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 200 > $O/profiles/r5_${w}_bench.json 2> $O/${w}.err
done
python3 bench.py > $O/profiles/r5_default_bench_line.json 2> $O/default.err
tail -20 $O/collect.txt
