#!/bin/bash
# the whole GPU suite, logged
R=$(pwd); O=$R/gpurun_out/r5_tests; mkdir -p $O
N=${1:-1}
python3 -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu_$N.txt
cat $O/pytest_gpu_$N.txt
