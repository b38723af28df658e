#!/bin/bash
# frame timeline of one workload:  bash scripts/run_timeline_wl.sh <workload> <outfile> [marker]
WL=$1; OUT=$2; MARK=${3:-blend}; R=$(pwd); mkdir -p $(dirname $R/$OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlw
rocprofv3 --kernel-trace -d /tmp/tlw --output-format csv -- python3 $R/bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tlw.out 2> /tmp/tlw.err
T=$(ls /tmp/tlw/*/*kernel_trace.csv | head -1)
python3 $R/scripts/frame_timeline.py $T --marker $MARK > $R/$OUT 2>&1
tail -1 /tmp/tlw.out | cut -c1-200
