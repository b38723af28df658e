#!/bin/bash
# A/B: FeatureNet's top-down path on a second stream next to cascade level 0 (BMV_OVERLAP=1) vs one stream (default)
for o in 2 2 2 0 0; do
  BMV_OVERLAP=$o python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ov.json
  python - "$o" <<'PY'
import json, sys
d = json.loads(open('/tmp/ov.json').read())
print("BMV_OVERLAP", sys.argv[1], round(d["value"], 1), "eager", round(d["value_extra"]["sync_bracketed_eager"]["value"], 1), "pipelined", round(d["value_extra"]["pipelined_replay"]["value"], 1), {k: (round(v["avg_us"], 1), round(v["frac"], 3)) for k, v in d["roofline"]["levels"].items()})
PY
done
