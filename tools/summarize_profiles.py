#!/usr/bin/env python3
"""Copies the judged summaries of a tools/prof_cmd.sh run from gpurun_out/ into profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
for name in ("bench", "bench_serialised", "bench_config5", "bench_config3", "bench_u8", "bench_32768ch",
             "bench_driver_flags", "bench_rccl_world1"):
    src = os.path.join(go, "%s_%s.json" % (tag, name))
    if os.path.exists(src):
        line = [l for l in open(src).read().splitlines() if l.startswith('{"metric"')][-1]
        json.loads(line)
        open(os.path.join(pr, "%s_%s.json" % (tag, name)), "w").write(line + "\n")
# the bench line printed by the very runs rocprofv3 traced (its event-based FIR time belongs beside
# the kernel-stats csv of the same run)
for name, dst in (("stats", "bench_under_rocprofv3"), ("stats_ser", "bench_serialised_under_rocprofv3")):
    src = os.path.join(go, "%s_%s.log" % (tag, name))
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith('{"metric"')]
        if lines:
            json.loads(lines[-1])
            open(os.path.join(pr, "%s_%s.json" % (tag, dst)), "w").write(lines[-1] + "\n")
st2 = glob.glob(os.path.join(go, tag + "_stats_ser", "**", "*_kernel_stats.csv"), recursive=True)
if st2:
    shutil.copy(st2[0], os.path.join(pr, tag + "_kernel_stats_serialised.csv"))
st = glob.glob(os.path.join(go, tag + "_stats", "**", "*_kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(pr, tag + "_kernel_stats.csv"))
out = open(os.path.join(pr, tag + "_pmc_k_if_fir.txt"), "w")
out.write("rocprofv3 --pmc passes (separate runs, counters only) on k_if_fir<InF32,64,7,true,0,false> (one tile per workgroup: calls not overlapped), 8192 channels,\n"
          "bench.py --concurrency 0; mean per launch.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE\n"
          "counts 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): read bytes = 2*FETCH_SIZE*1024.\n")
# bench.py reads the traffic of the kernel form it times: the file is keyed by form
tpath = os.path.join(pr, "traffic_k_if_fir.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
if "bytes_per_launch" in traffic:  # the flat round-2 layout
    traffic = {}
vals = {}
for d in ("pmc1", "pmc2", "pmc3"):
    fs = glob.glob(os.path.join(go, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
        out.write("%-24s %14.6g  (n=%d)\n" % (k, vals[k], len(v)))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    rd, wr = 2 * vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    out.write("HBM traffic per launch: read %.4g B + write %.4g B = %.4g B (algorithmic 4.6854e9 B)\n"
              % (rd, wr, rd + wr))
    print("traffic per launch:", rd + wr)
    traffic["k_if_fir"] = {
        "kernel": "k_if_fir<InF32,64,7,true,0,false> (one tile per workgroup: calls not overlapped)",
        "channels": 8192, "samples_per_call": 65536,
        "bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                  "read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024",
        "source": "profiles/%s_pmc_k_if_fir.txt" % tag}
out.close()
# the two-tile form that runs when calls overlap (passes 4 and 5: default bench.py, regex k_if_fir_mt)
mt = {}
for d in ("pmc4", "pmc5"):
    fs = glob.glob(os.path.join(go, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    if fs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            mt[k] = (sum(v) / len(v), len(v))
if "FETCH_SIZE" in mt and "WRITE_SIZE" in mt:
    rd, wr = 2 * mt["FETCH_SIZE"][0] * 1024, mt["WRITE_SIZE"][0] * 1024
    with open(os.path.join(pr, tag + "_pmc_k_if_fir.txt"), "a") as out2:
        out2.write("\nThe two-tile form that runs when calls overlap (k_if_fir_mt<InF32,7,2>, default bench.py, the same\n"
                   "separate --pmc passes, n=%d launches): read %.4g B + write %.4g B = %.4g B per launch = %.3f x algorithmic.\n"
                   % (mt["FETCH_SIZE"][1], rd, wr, rd + wr, (rd + wr) / 4.6854e9))
    traffic["k_if_fir_mt"] = {
        "kernel": "k_if_fir_mt<InF32,7,2> (two tiles per workgroup: overlapped calls beside the whole-CU "
                  "serial stage, the default bench.py run)",
        "channels": 8192, "samples_per_call": 65536,
        "bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the default (overlapped) "
                  "bench.py, read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024",
        "source": "profiles/%s_pmc_k_if_fir.txt (last paragraph)" % tag}
if traffic:
    json.dump(traffic, open(tpath, "w"), indent=1)
print(open(os.path.join(pr, tag + "_pmc_k_if_fir.txt")).read())
