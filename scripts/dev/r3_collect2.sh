#!/bin/bash
set -u
O=gpurun_out/r3_profiles2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
PFX=r3 timeout 2400 bash scripts/collect_profiles.sh $O c2 c1 c3 c4 c5 ft > $O/collect.log 2>&1; echo "collect rc=$?"
ls $O | head -40
