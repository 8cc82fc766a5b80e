#!/usr/bin/env python3
"""Print the key figures of bench.py JSON lines read from stdin (one line per run)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(tag, "value", d["value"], "ms/step", d["ms_per_step"], "| fir in-pipe ms", r["avg_ms"], "GB/s", r["achieved"],
          "frac", r["frac"], "TF", r["valu_tflops_nofma"], "| alone", r.get("alone", {}).get("avg_ms"))
    if d.get("stage_ms"):
        print("   stages:", " ".join("%s=%.3f" % (k, v) for k, v in d["stage_ms"].items()))
