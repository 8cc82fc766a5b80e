#!/usr/bin/env python3
"""Per call of a traced bench.py run (rocprofv3 --kernel-trace): when its IF FIR, serial stage and
audio tail started and ended, in ms relative to the first call shown -- the fill and drain of the
pipeline around a short timed region.  usage: trace_steps.py <rocprof dir> [first_call] [n_calls]"""
import csv
import glob
import sys

path = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
fir, ser, tail, hb, rs, pll = [], [], [], [], [], []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_if_fir" in n:
        fir.append((s, e))
    elif "k_demod_serial" in n:
        ser.append((s, e))
    elif "k_audio_tail" in n:
        tail.append((s, e))
    elif "k_halfband" in n or "k_rds_decim" in n:
        hb.append((s, e))
    elif "k_resample" in n:
        rs.append((s, e))
    elif "k_rds_pll" in n or "k_rds_light" in n:
        pll.append((s, e))
for v in (fir, ser, tail, hb, rs, pll):
    v.sort()
n = min(len(fir), len(ser), len(tail))
hb_per = max(1, round(len(hb) / max(1, len(ser))))  # half-band launches per call: the first one = heavy start
t0 = fir[first][0]
print("call   fir_start fir_end  ser_start ser_end  heavy_start rs_end  light_start tail_end   ser_gap  (ms)")
prev_end = None
for k in range(first, min(n, first + count)):
    gap = (ser[k][0] - prev_end) / 1e6 if prev_end is not None else 0.0
    prev_end = ser[k][1]
    h0 = hb[k * hb_per][0] if k * hb_per < len(hb) else t0
    r1 = rs[k][1] if k < len(rs) else t0
    l0 = pll[k][0] if k < len(pll) else t0
    print("%4d  %9.3f %8.3f  %9.3f %8.3f  %10.3f %7.3f  %10.3f %8.3f  %8.3f" % (
        k, (fir[k][0] - t0) / 1e6, (fir[k][1] - t0) / 1e6, (ser[k][0] - t0) / 1e6,
        (ser[k][1] - t0) / 1e6, (h0 - t0) / 1e6, (r1 - t0) / 1e6, (l0 - t0) / 1e6,
        (tail[k][1] - t0) / 1e6, gap))
