#!/bin/bash
# Per-round evidence (PFX=r6 by default) for profiles/: per workload a bench line (bench.py, cpu_baseline included) and the rocprofv3
# --kernel-trace summary of the same command reduced to its steady-state steps (scripts/rocprof_steady.py).
# Run on the MI355X box from the repo root:  bash scripts/collect_profiles.sh <outdir> [workload ...]
set -u
OUT=${1:-gpurun_out/r6_profiles}; shift || true
PFX=${PFX:-r6}
RM=${RENDER_MARKER:-render_pc_kernel}
R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name workload marker markers-per-step steps warmup extra...
  local name=$1 wl=$2 marker=$3 mps=$4 steps=$5 warm=$6; shift 6
  python3 $R/bench.py --workload $wl --steps $steps --warmup $warm --cpu-baseline "$@" > $R/$OUT/${name}_bench.json 2> $R/$OUT/${name}_bench.err
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name --output-format csv -- python3 $R/bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline "$@" > /tmp/prof_$name.out 2> /tmp/prof_$name.err
  local T=$(ls /tmp/prof_$name/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/rocprof_steady.py $T --marker "$marker" --markers-per-step $mps --steps $((steps - 1)) --out $R/$OUT/${name}_steady_kernel_stats.csv > $R/$OUT/${name}_steady.txt 2>&1
  head -3 $R/$OUT/${name}_steady.txt
  rm -rf /tmp/prof_$name
}
WL=${*:-"c2 c1 c3 c4 c5"}
for w in $WL; do
  case $w in
    c2) run ${PFX}_config2_enerf512x640 enerf_512x640_3src_64planes "$RM" 1 200 5 ;;
    c1) run ${PFX}_config1_enerf256x320 enerf_256x320_3src_32planes "$RM" 1 100 3 ;;
    c3) run ${PFX}_config3_enerf_ours480x736_k4 enerf_ours_480x736_6src_k4 "blend_wave_kernel" 1 60 3 ;;
    c4) run ${PFX}_config4_mvsnerf_ours_128planes_k4 mvsnerf_ours_224x352_128planes_k4 "blend_wave_kernel" 1 4 2 ;;
    c5) run ${PFX}_config5_enerf_ours_ft480x736_k4 enerf_ours_ft_480x736_6src_k4 "blend_bwd_kernel" 2 6 4 ;;
    ft) run ${PFX}_enerf_ft512x640 enerf_ft_512x640_3src "nerf_mlp_bwd_kernel<8, 3>" 1 16 6 ;;
  esac
done
ls -la $R/$OUT
