#!/bin/bash
# Texture-addresser / queue counters of the sweep kernels on the frame's own inputs.
#   bash scripts/pmc_sweep_ta.sh <outdir> <algos>
OUT=${1:-gpurun_out/pmc_sweep_ta}; ALGOS=${2:-4,200,202}; R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_ta_$i
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_ta_$i --output-format csv -- python3 $R/scripts/prof_sweep_once.py $ALGOS 3 > /tmp/pmc_ta_$i.out 2>&1
  python3 $R/scripts/pmc_summarize.py /tmp/pmc_ta_$i | tee $R/$OUT/sweep_ta_pass$i.txt
done
