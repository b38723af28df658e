#!/bin/bash
# the renderer's lookups from records written by the producing convolutions' epilogues: none / image records / image +
# volume records
for i in 1 2 3; do
  BMV_LOOKUP_RECORDS=0 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py planar | grep -o "render_rays[^)]*)\|^[a-z+]* *[0-9.]* "
  BMV_VOLUME_RECORDS=0 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py image | grep -o "render_rays[^)]*)\|^[a-z+]* *[0-9.]* "
  python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 scripts/bench_line.py image+volume | grep -o "render_rays[^)]*)\|^[a-z+]* *[0-9.]* "
done
