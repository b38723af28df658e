#!/bin/bash
# round-3 evidence: full GPU suite, bench lines + steady kernel stats for configs 1-5 and the plain fine-tune step,
# frame timeline, per-kernel MFMA / LDS counters, the training MLP micro-benchmark
set -u
O=gpurun_out/r3_profiles; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
PFX=r3 timeout 2400 bash scripts/collect_profiles.sh $O c2 c1 c3 c4 c5 ft > $O/collect.log 2>&1; echo "collect rc=$?"
timeout 900 bash scripts/run_timeline.sh $O/timeline > $O/timeline.log 2>&1; echo "timeline rc=$?"
timeout 900 bash scripts/pmc_frame.sh $O/pmc > $O/pmc.log 2>&1; echo "pmc rc=$?"
python scripts/bench_mvs_mlp_train.py 131072 > $O/mvs_mlp_train_bench.txt 2>&1
python scripts/bench_mvs_mlp_train.py 32768 >> $O/mvs_mlp_train_bench.txt 2>&1
grep forward $O/mvs_mlp_train_bench.txt
ls $O
