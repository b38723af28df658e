#!/bin/bash
# round-5: accuracy of the weight-gradient kernel by itself; kernel profile of the deterministic fine-tune step
R=$(pwd); O=$R/gpurun_out/r5n; mkdir -p $O
python3 tests/tools/wgrad_accuracy.py > $O/wgrad_accuracy.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export BMV_DETERMINISTIC=1
rm -rf /tmp/prof_det
rocprofv3 --kernel-trace --stats -d /tmp/prof_det --output-format csv -- python3 $R/bench.py --workload enerf_ft_512x640_3src --steps 8 --warmup 4 --no-cpu-baseline > $O/det_prof.out 2> $O/det_prof.err
T=$(ls /tmp/prof_det/*/*kernel_trace.csv | head -1)
python3 $R/scripts/rocprof_steady.py $T --marker "nerf_mlp_bwd_kernel<8, 3>" --markers-per-step 1 --steps 7 --out $O/det_ft_steady_kernel_stats.csv > $O/det_ft_steady.txt 2>&1
head -30 $O/det_ft_steady.txt; cat $O/wgrad_accuracy.txt
