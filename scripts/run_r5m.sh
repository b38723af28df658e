#!/bin/bash
# round-5 evidence batch: MFMA shape probe, cost of the deterministic scatter mode, per-workload bench + rocprof summaries
R=$(pwd); O=$R/gpurun_out/r5m; mkdir -p $O
$R/scripts/ubench/mfma_f32_shapes > $O/mfma_f32_shapes_ubench.txt 2>&1
for d in 0 1; do
  BMV_DETERMINISTIC=$d python3 bench.py --workload enerf_ft_512x640_3src --steps 16 --warmup 6 --no-cpu-baseline > $O/ft_det$d.json 2> $O/ft_det$d.err
  BMV_DETERMINISTIC=$d python3 bench.py --workload enerf_ours_ft_480x736_6src_k4 --steps 6 --warmup 4 --no-cpu-baseline > $O/c5_det$d.json 2> $O/c5_det$d.err
done
PFX=r5 bash scripts/collect_profiles.sh gpurun_out/r5m/profiles c2 c1 c3 c4 c5 ft > $O/collect.txt 2>&1
tail -30 $O/collect.txt
