#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -x -k "sweep" 2>&1 | tail -2
