#!/usr/bin/env python3
"""Random decoder geometries (IF rate, downsample, IF filter order, tuner table size, de-emphasis)
and random call sizes through the HIP path and the CPU oracle, bit for bit.  Exercises the launch
code's choices (tile size, window layout E = 0 / 1 / 2, load depth, power-of-two and '%' tuner
paths, hand-scheduled loops, short-block regimes) on shapes no fixed test names.
usage: fuzz_geometries.py [trials] [seed] [force]     (force = 1: the large-batch forms of resampler and
half-band chain -- k_resample_ring, k_halfband_chain -- wherever the geometry allows them)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tools import fmsig_py  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
FORCE = len(sys.argv) > 3 and sys.argv[3] == "1"
forced = 0
rng = np.random.default_rng(SEED)
pkg = load_package()
bad = 0
done = 0
for trial in range(T):
    fs = float(rng.choice([250e3, 400e3, 644e3, 900e3, 1.0e6, 1.2e6, 1.4e6, 1.8e6, 2.048e6, 2.4e6, 2.56e6,
                           2.88e6, 3.2e6, 10e6]))
    D = max(1, int(fs / 215e3))
    if rng.random() < 0.25 and D > 2:
        D += int(rng.integers(-1, 2))
    fb = fs / D
    if fb < 190e3 or fb >= 5.3e6:
        continue
    order = 0 if rng.random() < 0.4 else int(rng.integers(8, 1800))
    table = int(rng.choice([0, 0, 32, 64, 100, 128, 256]))
    us = bool(rng.random() < 0.3)
    kw = dict(if_filter_order=order, table_size=table, us_version=us)
    try:
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, **kw), 2)
    except pkg.FmdError as e:
        print("trial %d fs %.0f D %d order %d table %d: rejected at create (%s)" % (trial, fs, D, order, table, e))
        continue
    if FORCE:
        b.debug_set("halfband_chain", 1)
        try:
            b.debug_set("resampler", 1)
            forced += 1
        except pkg.FmdError:
            pass  # no form of the ring resampler fits this geometry's window
    o = oracle_py.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, **kw)
    p = fmsig_py.default_params(fs, noise_sigma=0.01, seed=100 + trial)
    nmin = b.min_samples()
    nmax = min(65536, (32768 - 52) * D)
    start = 0
    ok = True
    sizes = [nmax] + [int(rng.integers(nmin, nmax + 1)) for _ in range(5)] + [nmin, nmax]
    for k, n in enumerate(sizes):
        iq = fmsig_py.generate_f32(p, start, n)
        start += n
        r = o.process_stream(iq)
        a = b.process_host(np.stack([iq, iq]).view(np.complex64).reshape(2, n))
        if a[1].shape != r.shape or not np.array_equal(a[1].view(np.uint32), r.view(np.uint32)):
            ok = False
            print("MISMATCH trial %d fs %.0f D %d order %d table %d us %d call %d size %d"
                  % (trial, fs, D, order, table, us, k, n))
            break
    bad += 0 if ok else 1
    done += 1
    b.close()
print("fuzz: %d geometries (%d with the ring resampler forced), %d with a mismatch" % (done, forced, bad))
sys.exit(1 if bad else 0)
