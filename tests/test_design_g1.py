"""design-level vectors (Lanczos tables, Kaiser low-passes, biquads, tuner tables,
RDS matched filters and half-band chains).  The oracle must reproduce the committed set, and the
product's host-side constructors (csrc/fmd_design.hpp through the host-only C ABI entry points,
no GPU involved) must equal it bit for bit."""
import os

import numpy as np

from __graft_entry__ import ROOT, load_package
from tools.make_golden import design_vectors


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "design_g1.npz"))


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_oracle_reproduces_g1(oracle):
    g = _golden()
    v = design_vectors(oracle)
    assert len(v) == 18
    for k, a in v.items():
        assert _same_bits(a, g[k]), k
    for fs, D in ((2.4e6, 11), (1.0e6, 4)):
        dec = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
        key = "%d" % int(fs / D)
        assert _same_bits(dec.rds_mf_taps(), g["rds_mf_" + key])
        assert dec.rds_hb_lengths() == list(g["rds_hb_" + key])
        # the decoder's own filters are the G1 ones
        assert _same_bits(dec.audio_taps(), g["kaiser_audio_lpf_48k"])
        assert _same_bits(dec.rds_lpf_taps(), g["kaiser_rds_lpf_27272" if D == 11 else "kaiser_rds_lpf_31250"])
        assert _same_bits(dec.if_taps(), g["lanczos_88_0.054545"] if D == 11 else
                          oracle.design_lanczos(32, 0.6 / 4))
        assert _same_bits(dec.lut().view(np.float32), g["lut_64_10"])  # lrint(9.6) = 10
    # half-band chains [15, 23, 43] @218 kHz, [15, 19, 35] @250 kHz; 44 / 52 MF taps
    assert list(g["rds_hb_218181"]) == [15, 23, 43] and list(g["rds_hb_250000"]) == [15, 19, 35]
    assert g["rds_mf_218181"].size == 44 and g["rds_mf_250000"].size == 52


def test_product_host_design_equals_g1():
    pkg = load_package()
    g = _golden()
    v = design_vectors(pkg)
    for k, a in v.items():
        assert _same_bits(a, g[k]), k
    # Lanczos table: zero guard entries at both ends, unit DC gain (DownConvert.cpp:49-55)
    c = g["lanczos_88_0.054545"]
    assert c[0] == 0.0 and c[-1] == 0.0 and abs(float(c.astype(np.float64).sum()) - 1.0) < 1e-6
    # tuner table amplitude 2 (FmDecode.cpp:56), entry 0 = (2, 0)
    lut = g["lut_64_-7"].reshape(-1, 2)
    assert lut[0, 0] == 2.0 and lut[0, 1] == 0.0
    assert np.allclose(np.hypot(lut[:, 0], lut[:, 1]), 2.0, atol=1e-6)
