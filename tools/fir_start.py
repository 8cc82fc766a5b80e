#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: per steady-state period, when the IF FIR starts and ends
relative to the serial stage it runs beside, and when the heavy chain before it ended."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
               "fir" if "k_if_fir" in n else "ser" if "k_demod_serial" in n else "rs" if "k_resample" in n
               else "ring2" if "k_ring_fir4<HIP" in n else "tail" if "k_audio_tail" in n else "bits" if "k_rds_bits" in n else "o"))
ev.sort()
ser = [e for e in ev if e[2] == "ser"]
rows = []
for k in range(20, min(len(ser) - 2, 60)):
    s0, e0 = ser[k][0], ser[k][1]
    nxt = ser[k + 1][0]
    fir = [e for e in ev if e[2] == "fir" and s0 < e[0] < nxt]
    ring = [e for e in ev if e[2] == "ring2" and s0 < e[0] < nxt]
    tail = [e for e in ev if e[2] == "tail" and s0 < e[1] < nxt]
    bits = [e for e in ev if e[2] == "bits" and s0 < e[1] < nxt]
    if fir and ring:
        rows.append(((fir[0][0] - s0) / 1e3, (fir[0][1] - s0) / 1e3, (e0 - s0) / 1e3, (nxt - s0) / 1e3,
                     (ring[-1][1] - s0) / 1e3, (tail[-1][1] - s0) / 1e3 if tail else -1, (bits[-1][1] - s0) / 1e3 if bits else -1))
print("fir_start fir_end ser_end next_ser_start heavy_end(audio lpf) tail_end bits_end   (us from serial start), mean over %d periods" % len(rows))
print(" ".join("%8.1f" % (sum(r[i] for r in rows) / len(rows)) for i in range(7)))
for r in rows[:6]:
    print(" ".join("%8.1f" % x for x in r))
