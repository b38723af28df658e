#!/bin/bash
# HBM traffic of the two plane sweeps on the frame's own inputs: one rocprofv3 --pmc pass per counter
# (MI355X_MICROARCH.md: separate passes; FETCH_SIZE is doubled on gfx950 when summarised).
#   bash scripts/pmc_sweep.sh <outdir>
OUT=${1:-gpurun_out/pmc_sweep}; R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c --output-format csv -- python3 $R/scripts/prof_sweep_once.py 0,1 3 > /tmp/pmc_$c.out 2>&1
  python3 $R/scripts/pmc_summarize.py /tmp/pmc_$c | tee $R/$OUT/sweep_$c.txt
  cp $(ls /tmp/pmc_$c/*/*counter_collection.csv | head -1) $R/$OUT/sweep_frame_inputs_$c.csv
done
