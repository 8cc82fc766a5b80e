#!/usr/bin/env python3
"""Copies the judged summaries of a tools/prof_r4.sh run from gpurun_out/<tag>/ into profiles/<tag>_* and
derives profiles/traffic_per_call.json (what bench.py quotes as fabric_bytes_per_call) and the IF FIR's
traffic_k_if_fir.json from the counter passes.

    python tools/summarize_r4.py <tag>

FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side (fabric requests, Infinity Cache hits included);
on gfx950 FETCH_SIZE counts half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM), so read
bytes = 2 * FETCH_SIZE * 1024 for every kernel of this path (all of them read rows of >= 256 B per wave)."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for f in sorted(os.listdir(go)):
    src = os.path.join(go, f)
    if f.endswith(".json"):
        lines = [l for l in open(src).read().splitlines() if l.startswith('{"metric"')]
        if lines:
            json.loads(lines[-1])
            open(os.path.join(pr, "%s_%s" % (tag, f)), "w").write(lines[-1] + "\n")
    elif f in ("stats.log", "stats_ser.log"):
        lines = [l for l in open(src).read().splitlines() if l.startswith('{"metric"')]
        if lines:
            name = "bench_under_rocprofv3.json" if f == "stats.log" else "bench_serialised_under_rocprofv3.json"
            open(os.path.join(pr, "%s_%s" % (tag, name)), "w").write(lines[-1] + "\n")
    elif f in ("kernel_stats.csv", "kernel_stats_ser.csv", "pmc_all_kernels.txt", "pmc_k_if_fir_mt.txt"):
        dst = {"kernel_stats.csv": "kernel_stats.csv", "kernel_stats_ser.csv": "kernel_stats_serialised.csv"}.get(f, f)
        shutil.copy(src, os.path.join(pr, "%s_%s" % (tag, dst)))


def table(path):
    ctr, name = {}, None
    for line in open(path):
        if not line.strip():
            continue
        if not line.startswith(" "):
            name = line.strip()
            ctr.setdefault(name, {})
        else:
            k, v = line.split()[:2]
            ctr[name][k] = float(v)
    return ctr


pmc = os.path.join(pr, tag + "_pmc_all_kernels.txt")
stats = os.path.join(pr, tag + "_kernel_stats_serialised.csv")
if os.path.exists(pmc) and os.path.exists(stats):
    ctr = table(pmc)
    calls = {}
    for r in csv.DictReader(open(stats)):
        n = r["Name"].split("(")[0].replace("void ", "").replace("fmd::", "")[:34]
        calls[n] = calls.get(n, 0) + int(r["Calls"])
    ncall = max(v for k, v in calls.items() if k.startswith("k_demod_serial"))
    rows, total = [], 0.0
    for n, c in sorted(ctr.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or n not in calls:
            continue
        per_launch = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        per_call = per_launch * calls[n] / ncall
        rows.append({"kernel": n, "launches_per_call": round(calls[n] / ncall, 2), "read_bytes_per_launch": int(2 * c["FETCH_SIZE"] * 1024),
                     "write_bytes_per_launch": int(c["WRITE_SIZE"] * 1024), "bytes_per_call": int(per_call)})
        total += per_call
    rows.sort(key=lambda r: -r["bytes_per_call"])
    json.dump({"channels": 8192, "samples_per_call": 65536, "bytes_per_call": int(total), "kernels": rows,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the serialised bench, per kernel "
                         "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 * launches per call; kernels below 0.01 ms not listed",
               "source": "profiles/%s_pmc_all_kernels.txt" % tag},
              open(os.path.join(pr, "traffic_per_call.json"), "w"), indent=1)
    print("traffic per call: %.3f GB" % (total / 1e9))
    for r in rows[:12]:
        print("   %-36s %6.2f x  %8.1f MB" % (r["kernel"], r["launches_per_call"], r["bytes_per_call"] / 1e6))
    # the FIR's own file (bench.py's roofline.traffic), keyed by the kernel form
    tpath = os.path.join(pr, "traffic_k_if_fir.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for n, c in ctr.items():
        if n.startswith("k_if_fir<InF32, 64, 7") and "FETCH_SIZE" in c:
            traffic["k_if_fir"].update(bytes_per_launch=int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
                                       read_bytes=int(2 * c["FETCH_SIZE"] * 1024), write_bytes=int(c["WRITE_SIZE"] * 1024),
                                       source="profiles/%s_pmc_all_kernels.txt" % tag)
    mt = os.path.join(pr, tag + "_pmc_k_if_fir_mt.txt")
    if os.path.exists(mt):
        for n, c in table(mt).items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                traffic["k_if_fir_mt"].update(bytes_per_launch=int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
                                              read_bytes=int(2 * c["FETCH_SIZE"] * 1024), write_bytes=int(c["WRITE_SIZE"] * 1024),
                                              source="profiles/%s_pmc_k_if_fir_mt.txt" % tag,
                                              kernel=n.strip() + " (two tiles per workgroup: overlapped calls beside the "
                                                     "whole-CU serial stage, the default bench.py run; k_if_fir_mt3 = "
                                                     "two outputs per lane, tiles of 128 outputs)")
    json.dump(traffic, open(tpath, "w"), indent=1)
