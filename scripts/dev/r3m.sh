#!/bin/bash
set -u
O=gpurun_out/r3m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mvs.py -x -q > $O/mvs.log 2>&1; echo "mvs rc=$?"; tail -5 $O/mvs.log | cut -c1-250
python scripts/bench_mvs_mlp_train.py 131072 2>&1 | grep forward; python scripts/bench_mvs_mlp_train.py 32768 2>&1 | grep forward
bash scripts/dev/r3n.sh 2>&1 | grep "rows_\|mvs_\|==" 
