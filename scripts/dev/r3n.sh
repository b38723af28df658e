#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40
