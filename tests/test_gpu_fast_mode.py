"""The opt-in shuffle-reduced IF FIR (fmd_params::fir_reduction = FMD_FIR_SHUFFLE_PARITY_WAIVED): BASELINE's north star words
the per-tap reduction as wavefront shuffles; that changes the order of the float additions, so it
cannot be bit-identical to the reference's sequential sum (DownConvert.cpp:117-121).  This test
measures how far it lands from the parity mode on BASELINE config 2 (one stereo + RDS channel,
2.4 MS/s, 220 blocks = 6 s): 1.17e-5 RMS, worst block 3.46e-5 -- above the north star's own 1e-5
gate, which is why the mode has to be asked for by a value that says parity is waived (a plain 1 is
refused) and why every reported figure uses the sequential sum."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def test_shuffle_reduction_distance_from_parity_mode(fmsig, capsys):
    pkg = load_package()
    fs, D, nblk = 2.4e6, 11, 220
    p = fmsig.default_params(fs, noise_sigma=0.005)
    exact = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    with pytest.raises(pkg.FmdError):  # a plain 1 does not get the mode
        pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, fir_reduction=1), 1)
    fast = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D,
                                     fir_reduction=pkg.FIR_SHUFFLE_PARITY_WAIVED), 1)
    exact.enable_taps()
    fast.enable_taps()
    se = sa = 0.0
    n = 0
    worst_blk = 0.0
    fir_rel = 0.0
    ge, gf = [], []
    for blk in range(nblk):
        iq = fmsig.generate_f32(p, blk * N, N).view(np.complex64)
        a = exact.process_host(iq, shared=True)[0].astype(np.float64)
        f = fast.process_host(iq, shared=True)[0].astype(np.float64)
        assert a.shape == f.shape
        if blk == 0:
            de, df = exact.tap("demod"), fast.tap("demod")
            fir_rel = float(np.sqrt(np.mean(np.abs(de - df) ** 2)) / np.sqrt(np.mean(np.abs(de) ** 2)))
            assert 0.0 < fir_rel < 1e-6  # a few ulp: a different summation order, not a different filter
        if blk >= 1:
            d2 = float(np.sum((a - f) ** 2))
            se += d2
            sa += float(np.sum(a ** 2))
            n += a.size
            worst_blk = max(worst_blk, np.sqrt(d2 / a.size))
    rms = np.sqrt(se / n)
    with capsys.disabled():
        print("\nshuffle-reduced FIR vs parity mode over %d blocks: audio RMS difference %.3g "
              "(worst block %.3g, signal RMS %.3g), FIR output relative difference %.3g"
              % (nblk - 1, rms, worst_blk, np.sqrt(sa / n), fir_rel))
    # the float32 noise floor of the two feedback PLLs (5e-6 .. 1.2e-5 for a
    # 1-ulp input perturbation): what include/fmd.h documents for the mode, with a margin of a quarter
    assert 0.0 < rms < 1.5e-5 and worst_blk < 4.5e-5
    assert exact.sink.frames.get(0, []) == fast.sink.frames.get(0, [])  # RDS content survives
    assert exact.status().stereo_detected == fast.status().stereo_detected == 1
    exact.close()
    fast.close()


@pytest.mark.parametrize("sigma", [0.005, 0.01, 0.05], ids=["clean", "noisy", "very-noisy"])
def test_fused_multiply_add_distance_from_parity_mode(fmsig, capsys, sigma):
    """FMD_FIR_FMA_PARITY_WAIVED: the reference's tap order with every multiply-add fused (v_pk_fma_f32) in the IF FIR
    (cDownsampleFilter::Process, DownConvert.cpp:117-121) and the two fractional resamplers (:203-232) -- what
    `-march=native` does to the reference itself.  Distance from the parity mode on BASELINE
    config 2's signal, clean and noisy, 6 s each; the numbers are docs/MEASUREMENTS.md's "price of bit-exactness"
    table.  A plain 2 does not get the mode."""
    pkg = load_package()
    fs, D, nblk = 2.4e6, 11, 220
    p = fmsig.default_params(fs, noise_sigma=sigma)
    exact = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    with pytest.raises(pkg.FmdError):
        pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, fir_reduction=2), 1)
    with pytest.raises(pkg.FmdError):  # another geometry: refused, not silently run in the parity mode
        pkg.Batch(pkg.make_params(1.0e6, -0.15e6, 48000.0, 15000.0, 4, fir_reduction=pkg.FIR_FMA_PARITY_WAIVED), 1)
    fast = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D,
                                     fir_reduction=pkg.FIR_FMA_PARITY_WAIVED), 1)
    exact.enable_taps()
    fast.enable_taps()
    se = sa = 0.0
    n = 0
    worst_blk = fir_rel = rs_rel = 0.0
    for blk in range(nblk):
        iq = fmsig.generate_f32(p, blk * N, N).view(np.complex64)
        a = exact.process_host(iq, shared=True)[0].astype(np.float64)
        f = fast.process_host(iq, shared=True)[0].astype(np.float64)
        assert a.shape == f.shape
        if blk == 0:
            de, df = exact.tap("demod"), fast.tap("demod")
            fir_rel = float(np.sqrt(np.mean(np.abs(de - df) ** 2)) / np.sqrt(np.mean(np.abs(de) ** 2)))
            assert 0.0 < fir_rel < 1e-6  # one rounding per tap instead of two: not a different filter
            re, rf = exact.tap("mono_rs"), fast.tap("mono_rs")
            rs_rel = float(np.sqrt(np.mean((re - rf) ** 2)) / np.sqrt(np.mean(re ** 2)))
        if blk >= 1:
            d2 = float(np.sum((a - f) ** 2))
            se += d2
            sa += float(np.sum(a ** 2))
            n += a.size
            worst_blk = max(worst_blk, np.sqrt(d2 / a.size))
    rms = np.sqrt(se / n)
    same_rds = exact.sink.frames.get(0, []) == fast.sink.frames.get(0, [])
    with capsys.disabled():
        print("\nfused multiply-add vs parity mode, noise sigma %g, %d blocks: audio RMS difference %.3g (worst block "
              "%.3g, signal RMS %.3g), IF FIR output relative difference %.3g, resampler output %.3g, UECP frames %s "
              "(%d)" % (sigma, nblk - 1, rms, worst_blk, np.sqrt(sa / n), fir_rel, rs_rel,
                        "equal" if same_rds else "DIFFER", len(exact.sink.frames.get(0, []))))
    # the float32 noise floor of the two feedback PLLs (5e-6 .. 1.2e-5 for a 1-ulp input
    # perturbation), with a margin
    assert 0.0 < rms < 3e-5 and worst_blk < 1e-4
    assert same_rds and exact.status().stereo_detected == fast.status().stereo_detected
    exact.close()
    fast.close()
