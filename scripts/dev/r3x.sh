#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for pr in 0 5; do
  rm -rf /tmp/tl_$pr
  BMV_CONV_PAIR_ROWS=$pr rocprofv3 --kernel-trace -d /tmp/tl_$pr --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > /tmp/tl_$pr.out 2> /tmp/tl_$pr.err
  T=$(ls /tmp/tl_$pr/*/*kernel_trace.csv | head -1)
  echo "== pair_rows=$pr"; python3 $R/scripts/frame_timeline.py $T | grep "conv_mfma_kernel<3, 3, 1, 1, [48], 1, true\|frame span"
  tail -1 /tmp/tl_$pr.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_extra'].get('host_batch_sync',{}).get('value'), d['value_extra']['host_batch_sync_eager']['value'])"
done
