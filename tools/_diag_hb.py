import sys, numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
from tools import fmsig_py
pkg = load_package()
fs, D, N = 2.4e6, 11, 65536
p = fmsig_py.default_params(fs, noise_sigma=0.01, seed=41)
par = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D)
f, pl = pkg.Batch(par, 1), pkg.Batch(par, 1)
f.debug_set("halfband_chain", 1); pl.debug_set("halfband_chain", 0)
f.enable_taps(); pl.enable_taps()
pos = 0
for blk, n in enumerate([N] * 4 + [40000, 3000]):
    iq = fmsig_py.generate_f32(p, pos, n); pos += n
    f.process_host(iq.view(np.complex64), shared=True); pl.process_host(iq.view(np.complex64), shared=True)
    a, b = f.tap("rds_lpf").view(np.float32), pl.tap("rds_lpf").view(np.float32)
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    print("blk", blk, "n", n, "len", a.size, "mismatches", bad.size, "first", bad[:12] // 2, "maxdiff", float(np.abs(a - b).max()) if a.size == b.size else None)
