#!/bin/bash
# variance stores of the windowed sweep with / without the non-temporal bit, in the frame (sweep events + headline)
R=$(pwd)
for aux in 0 2 0 2; do
  rm -f $R/boostmvsnerfs_amd/csrc/sweep_win.o
  BMV_WIN_DEFS="-DBMV_WIN_STORE_AUX=$aux" python -m boostmvsnerfs_amd.build > /tmp/b.log 2>&1
  for i in 1 2; do python3 bench.py --no-cpu-baseline 2>/dev/null | python3 scripts/bench_line.py aux=$aux | cut -c1-400; done
done
rm -f $R/boostmvsnerfs_amd/csrc/sweep_win.o
python -m boostmvsnerfs_amd.build > /tmp/b.log 2>&1
