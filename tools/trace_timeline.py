#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel mean duration and the timeline of one
steady-state step (start/end of every kernel relative to that step's k_demod_serial start)."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
dur = defaultdict(list)
for s, e, n in ev:
    dur[n].append(e - s)
print("%-40s %6s %10s" % ("kernel", "calls", "mean_us"))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-40s %6d %10.1f" % (n[:40], len(v), sum(v) / len(v) / 1e3))
ser = [(s, e) for s, e, n in ev if n.startswith("k_demod_serial<2")]  # the overlapped phase's form
if len(ser) <= 6:
    ser = [(s, e) for s, e, n in ev if n.startswith("k_demod_serial")]
if len(ser) > 6:
    per = [(ser[i + 1][0] - ser[i][0]) / 1e3 for i in range(len(ser) - 1)]
    print("k_demod_serial start-to-start (us):", [round(x) for x in per[-8:]])
    t0 = ser[-3][0]
    t1 = ser[-2][0]
    print("timeline of one step (us relative to its k_demod_serial start):")
    for s, e, n in ev:
        if t0 - 1500e3 <= s < t1:
            print("  %9.1f -> %9.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:50]))
