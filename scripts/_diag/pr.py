#!/usr/bin/env python3
"""Print value / ms_per_step / roofline kernel of bench JSON lines.  usage: pr.py file.json [...]"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("mean_launch_ms"))
    except Exception as e:
        print(f, "unreadable:", repr(e), open(f.replace(".json", ".err")).read()[-600:] if f.endswith(".json") else "")
