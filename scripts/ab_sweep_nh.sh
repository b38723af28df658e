#!/bin/bash
# A/B: level-0 windowed sweep with the two channel halves in one workgroup (BMV_SWEEP_WIN_NH=2) vs two workgroups (default)
for nh in 1 2 1 2; do
  BMV_SWEEP_WIN_NH=$nh python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/nh.json
  python - "$nh" <<'PY'
import json, sys
d = json.loads(open('/tmp/nh.json').read())
print("NH", sys.argv[1], round(d["value"], 1), {k: (round(v["avg_us"], 2), round(v["frac"], 3)) for k, v in d["roofline"]["levels"].items()})
PY
done
