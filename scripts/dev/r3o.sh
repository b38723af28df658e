#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_framegraph.py tests/test_gpu_boost.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -30
