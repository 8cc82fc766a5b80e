#!/usr/bin/env python3
"""Mean PMC counter value per launch and kernel from the passes of tools/pmc_heavy.sh."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "")[:34]
        per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (name, _, ctr), v in per.items():
        acc[name][ctr].append(v)
for name in sorted(acc):
    print(name)
    for ctr in sorted(acc[name]):
        v = acc[name][ctr]
        print("   %-34s %14.4g  (n=%d)" % (ctr, sum(v) / len(v), len(v)))
