#!/usr/bin/env python3
"""Joules per call, part by part (GPU box only):  python tools/power_by_stage.py [--seconds 4] > out.txt

Runs the headline workload (8192 channels, 2.4 MS/s, calls overlapped) with only ONE part of a call launched
at a time (`fmd_batch_debug_set "stage_mask"`: IF stage / serial stage / half-band chain / resampler / light
part's RDS half / audio half -- the other kernels are left out, every event is still recorded, results are
wrong by construction) while a thread samples `rocm-smi --showpower`.  For every part: calls per second at full
rate, package power, joules per call = power x time per call, and the same above the idle draw.  The last line is
the whole pipeline.  What it is for: docs/MEASUREMENTS.md's energy table -- which part of a call the joules go to
when the package sits at its power limit.  Test / measurement infrastructure, not part of the product."""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

# (GPU_MAX_HW_QUEUES left at HIP's default, like bench.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=5).stdout
                w = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                if w:
                    self.samples.append((time.perf_counter(), float(w.group(1)), int(c.group(1)) if c else 0))
            except Exception:
                pass
            time.sleep(0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--channels", type=int, default=8192)
    ap.add_argument("--fir-variants", action="store_true")
    ap.add_argument("--alt-forms", action="store_true")
    ap.add_argument("--fma", action="store_true",
                    help="the parity-waived fused multiply-add form (FMD_FIR_FMA_PARITY_WAIVED) instead of the parity mode")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"])
    args = ap.parse_args()
    import numpy as np  # noqa: F401
    import torch
    from __graft_entry__ import load_package
    from tools import fmsig_py
    pkg = load_package()
    C, N, FS, D, RING, LAG, NBUF = args.channels, 65536, 2.4e6, 11, 4, 3, 6
    dev = torch.device("cuda", 0)
    gen = fmsig_py.DeviceGenerator([fmsig_py.channel_params(FS, c) for c in range(C)], dev)
    u8 = args.input == "u8"
    iq = torch.empty((RING, C, N, 2), dtype=torch.uint8 if u8 else torch.float32, device=dev)
    for r in range(RING):
        gen.generate(iq[r], r * N, N)
    b = pkg.Batch(pkg.make_params(FS, -0.15 * FS, 48000.0, 15000.0, D,
                                  fir_reduction=pkg.FIR_FMA_PARITY_WAIVED if args.fma else 0), C, record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device=dev) for _ in range(NBUF)]
    st = torch.cuda.current_stream().cuda_stream

    def run(calls):
        for i in range(calls):
            b.process_device(iq[i % RING].data_ptr(), N, N, audio[i % NBUF].data_ptr(), a_stride, st, u8=u8)
            if i >= LAG:
                b.wait(stream=st, lag=LAG)
                b.collect_rds_array(cap=4 * C, stream=st, lag=LAG)
        b.wait(stream=st)
        b.collect_rds_array(cap=4 * C, stream=st)
        b.take_rds_lost()
        torch.cuda.synchronize()

    run(12)
    time.sleep(1.0)
    smp = Sampler()
    smp.start()
    time.sleep(1.5)
    idle = [w for _, w, _ in smp.samples]
    p_idle = sum(idle) / max(1, len(idle))
    print("idle: %.0f W (%d samples)" % (p_idle, len(idle)))
    parts = [("IF stage (tuner + FIR + level)", 1), ("serial stage", 2), ("half-band chain", 4), ("resampler", 8),
             ("light part, RDS half", 16), ("light part, audio half", 32), ("heavy part (chain + resampler)", 12),
             ("light part, both halves", 48), ("IF stage + serial stage", 3), ("whole call", 63)]
    rows = []
    for name, mask in parts:
        b.debug_set("stage_mask", mask)
        t0 = time.perf_counter()
        run(40)
        per = (time.perf_counter() - t0) / 40
        calls = max(200, int(args.seconds / per))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(calls)
        t1 = time.perf_counter()
        dt = (t1 - t0) / calls
        lo, hi = t0 + 0.25 * (t1 - t0), t1 - 0.1 * (t1 - t0)
        ws = [(w, c) for t, w, c in smp.samples if lo <= t <= hi]
        p = sum(w for w, _ in ws) / max(1, len(ws))
        clk = sum(c for _, c in ws) / max(1, len(ws))
        rows.append({"part": name, "mask": mask, "calls": calls, "ms_per_call": round(dt * 1e3, 4), "watts": round(p, 1),
                     "mhz": round(clk), "joules_per_call": round(p * dt, 4),
                     "joules_above_idle": round((p - p_idle) * dt, 4), "samples": len(ws)})
        print("%-34s mask %2d  %6d calls  %.4f ms/call  %6.0f W  %4.0f MHz  %.3f J/call  (%.3f above idle)  [%d samples]" % (
            name, mask, calls, dt * 1e3, p, clk, p * dt, (p - p_idle) * dt, len(ws)))
        time.sleep(0.5)
    if args.fir_variants:  # the IF stage alone in its other forms: what its joules are made of
        for name, keys in (("IF stage, two outputs per lane (shipped)", {}),
                           ("IF stage, one output per lane (k_if_fir_mt)", {"fir_ro": 1}),
                           ("IF stage, three outputs per lane", {"fir_ro": 3}),
                           ("IF stage, one tile per workgroup (k_if_fir)", {"fir_nt": 1}),
                           ("IF stage, two outputs per lane (shipped)", {})):
            b.debug_set("fir_ro", keys.get("fir_ro", 2))
            b.debug_set("fir_nt", keys.get("fir_nt", 0))
            b.debug_set("stage_mask", 1)
            run(40)
            calls = int(args.seconds / 0.95e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(calls)
            t1 = time.perf_counter()
            dt = (t1 - t0) / calls
            lo, hi = t0 + 0.25 * (t1 - t0), t1 - 0.1 * (t1 - t0)
            ws = [(w, c) for t, w, c in smp.samples if lo <= t <= hi]
            p = sum(w for w, _ in ws) / max(1, len(ws))
            clk = sum(c for _, c in ws) / max(1, len(ws))
            print("%-48s %.4f ms/call  %6.0f W  %4.0f MHz  %.3f J/call  (%.3f above idle)" % (
                name, dt * 1e3, p, clk, p * dt, (p - p_idle) * dt))
            time.sleep(0.5)
    if args.alt_forms:  # the other forms of the heavy kernels and of the serial stage, in joules
        def measure2(name, mask, keys, undo):
            for k_, v_ in keys.items():
                b.debug_set(k_, v_)
            b.debug_set("stage_mask", mask)
            t0 = time.perf_counter()
            run(40)
            per = (time.perf_counter() - t0) / 40
            calls = max(200, int(args.seconds / per))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(calls)
            t1 = time.perf_counter()
            dt = (t1 - t0) / calls
            lo, hi = t0 + 0.25 * (t1 - t0), t1 - 0.1 * (t1 - t0)
            ws = [(w, c) for t, w, c in smp.samples if lo <= t <= hi]
            p = sum(w for w, _ in ws) / max(1, len(ws))
            clk = sum(c for _, c in ws) / max(1, len(ws))
            print("%-58s %.4f ms/call  %6.0f W  %4.0f MHz  %.3f J/call  (%.3f above idle)" % (
                name, dt * 1e3, p, clk, p * dt, (p - p_idle) * dt))
            for k_, v_ in undo.items():
                b.debug_set(k_, v_)
            time.sleep(0.5)
        measure2("resampler: LDS ring, 8 waves x 2 outputs (shipped)", 8, {}, {})
        measure2("resampler: LDS ring, 4 waves x 4 outputs", 8, {"rsr_form": 1}, {"rsr_form": 0})
        measure2("resampler: window per wave (round 3)", 8, {"resampler": 0}, {"resampler": -1})
        measure2("half-band chain: one kernel (shipped)", 4, {}, {})
        measure2("half-bands: a launch per stage (round 3)", 4, {"halfband_chain": 0}, {"halfband_chain": -1})
        measure2("serial stage: whole CUs (shipped)", 2, {}, {})
        measure2("serial stage: whole CUs, waves claim their SIMD", 2, {"serial_claim": 1}, {"serial_claim": 0})
        measure2("serial stage: shared form", 2, {"serial_exclusive": 0}, {"serial_exclusive": 1})
    b.debug_set("stage_mask", 63)
    smp.stop = True
    print(json.dumps({"idle_watts": round(p_idle, 1), "channels": C, "rows": rows}))
    b.close()


if __name__ == "__main__":
    main()
