#!/bin/bash
mkdir -p gpurun_out/r3d
python -m boostmvsnerfs_amd.build > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep" 2>&1 | grep -v "^  File\|^Extension" | tail -15 | tee gpurun_out/r3d/pytest.log
timeout 240 python scripts/tune_sweep_win.py --variants 0,17 --ring 0,1,2,3,4,5,6,7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/tune.log
for fl in 1 2 4 7; do echo "== flags $fl"; BMV_SWEEP_RING_FLAGS=$fl timeout 120 python scripts/tune_sweep_win.py --variants -1 --ring 0,1,2,3 2>&1 | grep "ring\|level"; done | tee gpurun_out/r3d/ablate.log
touch boostmvsnerfs_amd/csrc/sweep_ring.hip
BMV_RING_DEFS=-DBMV_RING_STAMPS python -m boostmvsnerfs_amd.build > /dev/null 2>&1
timeout 200 python scripts/stamps_ring.py 2 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d/stamps.log
