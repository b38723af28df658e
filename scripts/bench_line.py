#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line read from stdin:  python bench.py ... | python scripts/bench_line.py [label]"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.readlines()[-1])
k = d.get("kernels", {})
ks = {n: (round(v["avg_us"], 1), v["launches"], str(v.get("timed"))[:30]) for n, v in k.items()}
sm = (d.get("value_extra") or {}).get("step_ms")
if sm:
    label = label + " [step ms min %.3f med %.3f p90 %.3f max %.3f]" % (sm["min"], sm["median"], sm["p90"], sm["max"])
print(label.ljust(14), round(d["value"], 1), d["unit"], "|", d["config"].get("launch"), "|", ks, "| frac", (d.get("roofline") or {}).get("frac"))
