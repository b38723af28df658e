#!/bin/bash
# A/B: FPN top-down kernels at the allocator's occupancy (5 workgroups per CU) vs 6 / 8
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for wpe in 1 6 8; do
  rm -f $R/boostmvsnerfs_amd/csrc/conv.o
  (cd $R && BMV_CONV_DEFS="-DBMV_FPN_WPE=$wpe" python -m boostmvsnerfs_amd.build > /tmp/build_fpn$wpe.log 2>&1)
  echo "== FPN WPE $wpe"
  for i in 1 2; do python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  headline', round(d['value'],1))"; done
  rm -rf /tmp/prof_fp
  rocprofv3 --kernel-trace --stats -d /tmp/prof_fp --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 -c "
import csv,glob
for f in glob.glob('/tmp/prof_fp/*/*kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if 'fpn_topdown' in r[0]: print('  ', r[0][:50], r[1], round(float(r[3])/1000,1))
"
done
rm -f $R/boostmvsnerfs_amd/csrc/conv.o
(cd $R && python -m boostmvsnerfs_amd.build > /dev/null 2>&1)
