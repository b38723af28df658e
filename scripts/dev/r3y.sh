#!/bin/bash
# round-3 GPU batch: training-path changes + calibration prints
set -u
O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_conv.py -x -q > $O/train_tests.log 2>&1; echo "train tests rc=$?"; tail -5 $O/train_tests.log
timeout 600 python -m pytest tests/test_gpu_boost.py tests/test_gpu_mvs.py tests/test_gpu_configs34.py -x -q -s > $O/boost_mvs.log 2>&1; echo "boost/mvs rc=$?"; tail -3 $O/boost_mvs.log; grep -h "\[boost flips\]\|\[embedding\]" $O/boost_mvs.log | sort | uniq -c
for g in 1 0; do
  timeout 600 python bench.py --workload enerf_ours_ft_480x736_6src_k4 --steps 8 --warmup 4 --graph $g --no-cpu-baseline > $O/c5_graph$g.json 2> $O/c5_graph$g.err; echo "c5 graph=$g rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/c5_graph$g.json").read().strip().splitlines()[-1]); print("c5 graph=$g", d["value"], d["ms_per_step"], d["config"]["launch"])
except Exception as e: print("parse fail", e); print(open("$O/c5_graph$g.err").read()[-1500:])
PY
done
for g in 1 0; do
  timeout 600 python bench.py --workload enerf_ft_512x640_3src --steps 16 --warmup 6 --graph $g --no-cpu-baseline > $O/ft_graph$g.json 2> $O/ft_graph$g.err; echo "ft graph=$g rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/ft_graph$g.json").read().strip().splitlines()[-1]); print("ft graph=$g", d["value"], d["ms_per_step"], d["config"]["launch"])
except Exception as e: print("parse fail", e); print(open("$O/ft_graph$g.err").read()[-1500:])
PY
done
