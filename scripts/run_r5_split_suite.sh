#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -q -x -k "split" 2>&1 | tail -3
BMV_RENDER_SPLIT=1 python3 -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest_gpu_with_render_split.txt
tail -25 $O/pytest_gpu_with_render_split.txt
