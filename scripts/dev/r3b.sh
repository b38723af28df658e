#!/bin/bash
# dev: ring sweep -- which test crashes; ablations of the default variants
mkdir -p gpurun_out/r3b
for t in test_sweep_extreme_coordinates "test_sweep_kernels_agree_at_scale" test_sweep_windowed_ragged_shapes; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$t" 2>&1 | grep -v "^  File\|^Extension" | head -60 > gpurun_out/r3b/pytest_$t.log
  echo "== $t"; head -40 gpurun_out/r3b/pytest_$t.log
done
for fl in 0 1 2 4 3 6 7; do
  echo "== flags $fl"
  BMV_SWEEP_RING_FLAGS=$fl timeout 120 python scripts/tune_sweep_win.py --variants -1 --ring 0,2 2>&1 | grep "ring\|level"
done | tee gpurun_out/r3b/ablate.log
