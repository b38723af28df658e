#!/bin/bash
# A/B: convolution engine with the tuned occupancy targets (default) vs the allocator's own (-DBMV_CONV_WPE_TUNED=0).
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for tuned in 0 1; do
  rm -f $R/boostmvsnerfs_amd/csrc/conv.o
  (cd $R && BMV_CONV_DEFS="-DBMV_CONV_WPE_TUNED=$tuned" python -m boostmvsnerfs_amd.build > /tmp/build_conv$tuned.log 2>&1)
  echo "== TUNED $tuned"
  for i in 1 2; do python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  headline', round(d['value'],1), round(d['ms_per_step'],4))"; done
  rm -rf /tmp/prof_cw
  rocprofv3 --kernel-trace --stats -d /tmp/prof_cw --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 -c "
import csv,glob
for f in glob.glob('/tmp/prof_cw/*/*kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if 'conv_mfma_kernel<3, 3, 1, 1, 8, 1, true>' in r[0] or 'conv_mfma_kernel<3, 3, 1, 1, 4, 1, false>' in r[0] or 'conv_mfma_kernel<1, 3, 1, 1, 8, 0, true>' in r[0]:
            print('  ', r[0][:60], r[1], round(float(r[3])/1000,1))
"
done
rm -f $R/boostmvsnerfs_amd/csrc/conv.o
(cd $R && python -m boostmvsnerfs_amd.build > /dev/null 2>&1)
