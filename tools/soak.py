#!/usr/bin/env python3
"""Soak: many differently impaired stations (detuned, weak, noisy, over-deviated, silent, DC, clipping)
through the HIP batch path and through the CPU oracle, block by block, bit for bit.  Looks for
rare-path divergences (literal arctangent / phase-wrap fallbacks, PLL slips, RDS sync loss and
FEC) that the fixed test signals may never reach.  usage: soak.py [channels] [blocks] [seed] [force]
(force = 1: the large-batch forms of resampler and half-band chain, k_resample_ring / k_halfband_chain)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tools import fmsig_py  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NBLK = int(sys.argv[2]) if len(sys.argv) > 2 else 120
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = 65536
fs, D = 2.4e6, 11
rng = np.random.default_rng(SEED)
params = []
for c in range(C):
    kind = c % 8
    kw = dict(noise_sigma=float(rng.choice([0.0, 0.003, 0.02, 0.08, 0.25])), seed=5000 + 17 * c + SEED,
              pi=int(rng.integers(1, 65535)), ps="SOAK%04d" % c, f_left=float(rng.uniform(100, 9000)),
              f_right=float(rng.uniform(100, 14000)))
    if kind == 1:
        kw.update(f_offset=float(-0.15 * fs + rng.uniform(-40e3, 40e3)))  # detuned
    elif kind == 2:
        kw.update(amp=float(rng.uniform(0.002, 0.02)))  # weak
    elif kind == 3:
        kw.update(dev=float(rng.uniform(90e3, 140e3)))  # over-deviated: the PLL hits its clamps
    elif kind == 4:
        kw.update(amp=0.0, noise_sigma=float(rng.choice([0.0, 0.01])))  # silence / noise only
    elif kind == 5:
        kw.update(amp=0.99, a_rds=0.15)  # hot, strong RDS
    elif kind == 6:
        kw.update(a_pilot=float(rng.uniform(0.0, 0.03)), a_rds=float(rng.uniform(0.0, 0.02)))  # marginal pilot / RDS
    params.append(fmsig_py.default_params(fs, **kw))
pkg = load_package()
b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C)
if len(sys.argv) > 4 and sys.argv[4] == "1":
    b.debug_set("resampler", 1)
    b.debug_set("halfband_chain", 1)
refs = [oracle_py.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(C)]
bad = 0
sizes = [N] * NBLK
for k in range(3, NBLK, 11):
    sizes[k] = int(rng.integers(100, N))  # ragged calls in between
start = 0
for k, n in enumerate(sizes):
    iq = np.stack([fmsig_py.generate_f32(p, start, n) for p in params])
    start += n
    a = b.process_host(iq.view(np.complex64).reshape(C, n))
    for c in range(C):
        r = refs[c].process_stream(iq[c])
        if a[c].shape != r.shape or not np.array_equal(a[c].view(np.uint32), r.view(np.uint32)):
            bad += 1
            print("MISMATCH block %d channel %d (kind %d)" % (k, c, c % 8))
for c in range(C):
    so, sg = refs[c].status(), b.status(c)
    if ((so.stereo, so.rds_state) != (sg.stereo_detected, sg.rds_state)
            or np.float32(so.pilot_level) != np.float32(sg.pilot_level)):
        bad += 1
        print("STATUS MISMATCH channel", c)
    if b.sink.frames.get(c, []) != refs[c].uecp_frames():
        bad += 1
        print("UECP MISMATCH channel", c)
locked = sum(int(b.status(c).stereo_detected) for c in range(C))
groups = sum(len(r.rds_groups()) for r in refs)
print("soak: %d channels x %d calls, %d mismatches; %d channels stereo-locked, %d RDS groups in total"
      % (C, NBLK, bad, locked, groups))
sys.exit(1 if bad else 0)
