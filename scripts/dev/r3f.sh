#!/bin/bash
for defs in "-DBMV_RING_WPE3=3" "-DBMV_RING_WPE3=4" "-DBMV_RING_XW=1" "-DBMV_RING_XW=3"; do
touch boostmvsnerfs_amd/csrc/sweep_ring.hip
BMV_RING_DEFS="$defs" python -m boostmvsnerfs_amd.build > /dev/null 2>&1
echo "=== $defs"
BMV_SWEEP_RING_DEBUG=1 timeout 100 python scripts/tune_sweep_win.py --variants -1 --ring 0,1,2,3 2>&1 | grep "\[ring\]\|ring " | sort | uniq -c
done
