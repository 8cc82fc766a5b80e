"""The fractional resamplers as a stream over an LDS ring (k_resample_ring, what large batches run)
against the CPU oracle and against the window-per-wave form (k_resample), bit for bit.

The library picks the ring form by batch size; here it is forced (fmd_batch_debug_set "resampler")
so that one-channel batches go through it too -- with as many time segments as the call allows, so
that the segment borders, the ring's wrap and the first window of a segment are all exercised.
Reference: cDownsampleFilter::Process(real), fractional branch, /root/reference/src/DownConvert.cpp:195-233."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("fs,D,form,sizes", [
    (2.4e6, 11, 0, [N] * 6 + [40000, 1000, 65535, 222, 50001, N]),   # 8 waves x 2 outputs (what large batches run)
    (2.4e6, 11, 1, [N] * 6 + [40000, 1000, 65535, 222, 50001, N]),   # 4 waves x 4 outputs
    (1.0e6, 4, 0, [N] * 4 + [30000, 800, 65535, N]),                 # 250 taps: only 4 waves x 2 outputs fit
    (10e6, 46, 0, [N] * 5),                                          # config 5's baseband rate
])
def test_ring_form_stage_taps_bit_exact(oracle, fmsig, fs, D, form, sizes):
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=31)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    b.debug_set("rsr_form", form)
    b.debug_set("resampler", 1)  # an error where no form fits the geometry
    b.enable_taps()
    pos = 0
    for blk, n in enumerate(sizes):
        iq = fmsig.generate_f32(p, pos, n)
        pos += n
        a_ref = o.process_stream(iq)
        a_gpu = b.process_host(iq.view(np.complex64), shared=True)[0]
        taps = o.taps()
        for name in ("mono_rs", "stereo_rs"):
            g, r = b.tap(name), taps[name]
            assert g.shape == r.shape, (blk, name, g.shape, r.shape)
            assert _bits_equal(g.view(np.float32), r.view(np.float32)), (blk, n, name)
        assert _bits_equal(a_gpu, a_ref), (blk, n)
    b.close()


@pytest.mark.parametrize("C", [130, 1024])
def test_ring_form_equals_window_form(oracle, fmsig, C):
    """Many channels (C = 130: a ragged last group; 1024: what the library itself switches at is larger,
    the grid is the same): both forms on the same inputs, every channel's audio identical; a few
    channels also against the oracle."""
    pkg = load_package()
    fs, D = 2.4e6, 11
    base = [fmsig.default_params(fs, noise_sigma=0.01, seed=300 + k, f_left=300.0 + 211 * k)
            for k in range(3)]
    check = [0, C // 2, C - 1]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    par = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D)
    ring, win = pkg.Batch(par, C, record_callbacks=False), pkg.Batch(par, C, record_callbacks=False)
    ring.debug_set("resampler", 1)
    win.debug_set("resampler", 0)
    for blk in range(4):
        src = [fmsig.generate_f32(p, blk * N, N) for p in base]
        iq = np.stack([src[c % 3] for c in range(C)]).view(np.complex64).reshape(C, N)
        a_r, a_w = ring.process_host(iq), win.process_host(iq)
        assert _bits_equal(a_r, a_w), blk
        for c in check:
            assert _bits_equal(a_r[c], refs[c].process_stream(src[c % 3])), (blk, c)
    ring.close()
    win.close()


def test_ring_form_with_non_finite_samples(fmsig):
    """Zero taps meet rows outside an output's own window: exact for finite samples only.  A block with
    infinities and NaNs in it must come out of the ring form exactly as out of the window form (which
    has its own literal path), NaN for NaN."""
    pkg = load_package()
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=77)
    par = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D)
    ring, win = pkg.Batch(par, 2, record_callbacks=False), pkg.Batch(par, 2, record_callbacks=False)
    ring.debug_set("resampler", 1)
    win.debug_set("resampler", 0)
    for blk in range(5):
        iq = np.stack([fmsig.generate_f32(p, blk * N, N)] * 2).copy()
        if blk == 2:  # channel 1 only: channel 0 has to stay clean in both forms
            iq[1, 2 * 30000] = np.inf
            iq[1, 2 * 41000 + 1] = np.nan
        a_r = ring.process_host(iq.view(np.complex64).reshape(2, N))
        a_w = win.process_host(iq.view(np.complex64).reshape(2, N))
        same = (a_r.view(np.uint32) == a_w.view(np.uint32)) | (np.isnan(a_r) & np.isnan(a_w))
        assert same.all(), blk
        assert np.isfinite(a_r[0]).all()
    ring.close()
    win.close()
