#!/bin/bash
mkdir -p gpurun_out/r3e
BMV_RING_DEFS=-DBMV_RING_STAMPS python -m boostmvsnerfs_amd.build > /dev/null 2>&1
for cap in 416 384 320 256; do echo "=== cap $cap"; BMV_SWEEP_RING_CAP=$cap timeout 200 python scripts/stamps_ring.py 2 0 2>&1 | grep "==\|gaps\|lifetime\|eager"; done | tee gpurun_out/r3e/stamps.log
