#!/bin/bash
mkdir -p gpurun_out/r3k
run() {
  touch boostmvsnerfs_amd/csrc/sweep_zp.hip
  BMV_ZP_DEFS="$1" python -m boostmvsnerfs_amd.build 2>&1 | grep -i "error"
  echo "=== $1"
  timeout 200 python scripts/tune_sweep_win.py --variants -1 --zp 0,1,2,3 2>&1 | grep "zp\|level"
}
(run "-DBMV_ZP_TAPBUF=2"; run "-DBMV_ZP_TAPBUF=1"; run "-DBMV_ZP_PHOIST=1"; run "-fno-slp-vectorize"; run "-DBMV_ZP_TAPBUF=1 -fno-slp-vectorize -DBMV_ZP_PHOIST=1") 2>&1 | tee gpurun_out/r3k/variants.log
