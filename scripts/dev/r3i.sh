#!/bin/bash
mkdir -p gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep" 2>&1 | grep -v "^  File\|^Extension" | tail -5 | tee gpurun_out/r3i/pytest.log
timeout 300 python scripts/tune_sweep_win.py --variants 0,17 --zp 0,1,2,3,4,5,12,11,6,9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3i/tune.log
bash scripts/pmc_sweep_sq.sh gpurun_out/r3i 200,202 2>&1 | grep -A8 "zp_kernel"
