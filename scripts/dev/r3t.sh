#!/bin/bash
set -u
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_mvs.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for wl in enerf_ft_512x640_3src enerf_ours_ft_480x736_6src_k4; do
  timeout 600 python bench.py --workload $wl --steps 12 --warmup 6 --no-cpu-baseline > $O/$wl.json 2> $O/$wl.err; echo "$wl rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/$wl.json").read().strip().splitlines()[-1]); print("$wl", round(d["value"],2), round(d["ms_per_step"],2), "ms")
except Exception as e: print("parse fail", e); print(open("$O/$wl.err").read()[-1500:])
PY
done
