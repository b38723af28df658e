#!/bin/bash
mkdir -p gpurun_out/r3c
BMV_RING_DEFS=-DBMV_RING_STAMPS python -m boostmvsnerfs_amd.build > /dev/null 2>&1
(timeout 200 python scripts/stamps_ring.py 2 0; echo "=== no blend"; STAMP_EXTRA_FLAGS=2 timeout 200 python scripts/stamps_ring.py 2 0;  echo "=== no fill no blend"; STAMP_EXTRA_FLAGS=3 timeout 200 python scripts/stamps_ring.py 2 0) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c/stamps.log
