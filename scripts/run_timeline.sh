#!/bin/bash
# frame timeline of the headline workload, fork off and on:  bash scripts/run_timeline.sh <outdir>
OUT=${1:-gpurun_out/timeline}; R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for ov in 0 2; do
  rm -rf /tmp/tl_$ov
  BMV_OVERLAP=$ov rocprofv3 --kernel-trace -d /tmp/tl_$ov --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > /tmp/tl_$ov.out 2> /tmp/tl_$ov.err
  T=$(ls /tmp/tl_$ov/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/frame_timeline.py $T > $R/$OUT/frame_overlap$ov.txt 2>&1
  tail -1 /tmp/tl_$ov.out | cut -c1-200
done
cat $R/$OUT/frame_overlap0.txt
