#!/usr/bin/env python3
"""Per-kernel table from one round's counter passes: what bounds every kernel of the headline
workload (and the config-5 FIR) when it has the chip to itself.

    python tools/kernel_bounds.py profiles/<tag>_pmc_all_kernels.txt profiles/<tag>_kernel_stats_serialised.csv

Inputs: tools/pmc_table.py's per-launch means (rocprofv3 --pmc, one run per counter group, serialised
bench) and the kernel-trace stats of a serialised run.  SQ_*_CYCLES counters are in units of 4
cycles and summed over all waves / SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import re
import sys

pmc, stats = sys.argv[1], sys.argv[2]
ctr, name = {}, None
for line in open(pmc):
    if not line.startswith(" "):
        name = line.strip()
        ctr[name] = {}
    else:
        k, v = line.split()[:2]
        ctr[name][k] = float(v)
dur = {}
for r in csv.DictReader(open(stats)):
    n = r["Name"].split("(")[0].replace("void ", "").replace("fmd::", "")[:34]
    dur[n] = float(r["AverageNs"]) / 1e6
print("| kernel | ms alone | waves/SIMD | VALU active | waiting (mem/LDS/barrier) | issue stall | LDS array busy | "
      "FETCH+WRITE MB (raw) | GB/s (raw) | VALU instr/launch |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n, c in sorted(ctr.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" not in c or "SQ_WAVE_CYCLES" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    ms = dur.get(n)
    wc = c["SQ_WAVE_CYCLES"]
    mb = (c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e6
    t = ms if ms else cyc / 2.3e6
    print("| `%s` | %s | %.2f | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.0f | %.0f | %.3g |" % (
        n, ("%.3f" % ms) if ms else "(%.3f)" % (cyc / 2.3e6), wc * 4 / (cyc * 1024),
        100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc,
        100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc),
        mb, mb / t, c.get("SQ_INSTS_VALU", 0)))
