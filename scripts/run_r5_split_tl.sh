#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_split; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/tls_$v
  BMV_RENDER_SPLIT=$v rocprofv3 --kernel-trace -d /tmp/tls_$v --output-format csv -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/tls_$v.out 2> /tmp/tls_$v.err
  T=$(ls /tmp/tls_$v/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/frame_timeline.py $T > $O/timeline_split$v.txt 2>&1
  grep "frame span\|render_pc" $O/timeline_split$v.txt | tail -6
  python3 - "$T" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
r=[x for x in rows if 'render_pc_kernel' in x['Kernel_Name']]
f=[x for x in rows if 'frame_feed_ring' in x['Kernel_Name']]
ends=[int(x['End_Timestamp']) for x in r]; starts=[int(x['Start_Timestamp']) for x in f]
import statistics
# period between consecutive renderer ends in the timed region (last 20)
per=[(ends[i+1]-ends[i])/1e3 for i in range(len(ends)-21,len(ends)-1)]; print("kernel of the last frames:", r[-1]["Kernel_Name"][:60])
print('renderer dur us', statistics.median([(int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3 for x in r[-20:]]), 'frame period us (median of last 20)', statistics.median(per))
# gap between a frame's renderer end and the next frame's first kernel
P
done
