#!/bin/bash
# the renderer's image taps from 48-byte lookup records written by the fused smooth0 epilogue vs planar maps
for i in 1 2 3; do
  BMV_LOOKUP_RECORDS=0 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py planar | cut -c1-150
  BMV_LOOKUP_RECORDS=1 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py records | cut -c1-150
done
