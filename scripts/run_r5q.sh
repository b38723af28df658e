#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5q; mkdir -p $O
python3 scripts/ablate_conv_c4.py > $O/conv_c4_ablation.txt 2>&1
cat $O/conv_c4_ablation.txt
python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_properties.py -q -x 2>&1 | tail -3
