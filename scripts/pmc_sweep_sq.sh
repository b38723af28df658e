#!/bin/bash
# Instruction / LDS counters of the sweep kernels on the frame's own inputs (one rocprofv3 --pmc pass per group).
#   bash scripts/pmc_sweep_sq.sh <outdir> <algos, e.g. 4,200,202>
OUT=${1:-gpurun_out/pmc_sweep_sq}; ALGOS=${2:-4,200,202}; R=$(pwd); mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc_sq_$i
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_sq_$i --output-format csv -- python3 $R/scripts/prof_sweep_once.py $ALGOS 3 > /tmp/pmc_sq_$i.out 2>&1
  python3 $R/scripts/pmc_summarize.py /tmp/pmc_sq_$i | tee $R/$OUT/sweep_sq_pass$i.txt
done
