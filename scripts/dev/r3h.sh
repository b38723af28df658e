#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pq --output-format csv -- python3 $R/scripts/prof_sweep_once.py 4,200,202 2 > /tmp/pq.out 2>&1
grep -v "^[EWI]2026" /tmp/pq.out | tail -20
head -2 $(ls /tmp/pq/*/*counter_collection.csv | head -1)
python3 $R/scripts/pmc_summarize.py /tmp/pq
