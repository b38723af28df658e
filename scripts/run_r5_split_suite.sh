#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O
BMV_RENDER_SPLIT=1 python3 -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu_with_render_split.txt
tail -8 $O/pytest_gpu_with_render_split.txt
