#!/bin/bash
mkdir -p gpurun_out/r3g
python -m boostmvsnerfs_amd.build 2>&1 | grep -i "error" 
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep" 2>&1 | grep -v "^  File\|^Extension" | tail -15 | tee gpurun_out/r3g/pytest.log
timeout 300 python scripts/tune_sweep_win.py --variants 0,17 --zp 0,1,2,3,4,5,6,7,8,9,10,11,12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g/tune.log
for fl in 1 2 4 7; do echo "== flags $fl"; BMV_SWEEP_ZP_FLAGS=$fl timeout 120 python scripts/tune_sweep_win.py --variants -1 --zp 0,1,2,3 2>&1 | grep "zp\|level"; done | tee gpurun_out/r3g/ablate.log
