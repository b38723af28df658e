#!/bin/bash
# A/B: fused renderer at 2 vs 3 workgroups (= waves per SIMD) per CU.  Run on the MI355X box from the repo root.
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for wps in 2 3; do
  rm -f $R/boostmvsnerfs_amd/csrc/render.o
  (cd $R && BMV_RENDER_DEFS="-DBMV_RENDER_WPS=$wps" python -m boostmvsnerfs_amd.build > /tmp/build_$wps.log 2>&1; grep -c "render.hip" /tmp/build_$wps.log)
  echo "== WPS $wps"
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'])"
  rm -rf /tmp/prof_wps
  rocprofv3 --kernel-trace --stats -d /tmp/prof_wps --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 -c "import csv,glob; [print(r[0][:60], r[1], r[3]) for f in glob.glob('/tmp/prof_wps/*/*kernel_stats.csv') for r in csv.reader(open(f)) if 'render_rays' in r[0]]"
done
rm -f $R/boostmvsnerfs_amd/csrc/render.o
(cd $R && python -m boostmvsnerfs_amd.build > /dev/null 2>&1)
