#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for sp in auto 0; do
  rm -rf /tmp/tl_$sp
  BMV_CONV_SPLIT=$sp rocprofv3 --kernel-trace -d /tmp/tl_$sp --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > /tmp/tl_$sp.out 2> /tmp/tl_$sp.err
  T=$(ls /tmp/tl_$sp/*/*kernel_trace.csv | head -1)
  echo "== split=$sp"; python3 $R/scripts/frame_timeline.py $T | grep "conv3d_split\|conv_mfma_kernel<3, 3, 1, 1, [48], 1\|frame span\|depth_regress\|render_pc" | cut -c1-130
done
