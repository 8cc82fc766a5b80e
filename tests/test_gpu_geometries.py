"""Parity over other geometries the constructor accepts: odd / even / unit decimation, a
non-power-of-two tuner table, 75 us de-emphasis, channel counts that exercise lane padding and
both block->channel mappings, and the size limits of a call."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_unit_decimation_half_blocks(oracle, fmsig):
    """downsample = 1 (fs = 250 kHz): blocks of 32000 samples, the largest class of call the
    reference itself can process without overrunning its half-band buffers."""
    pkg = load_package()
    fs, D, n = 250e3, 1, 32000
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=11)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    for blk in range(6):
        iq = fmsig.generate_f32(p, blk * n, n)
        r = o.process_stream(iq)
        a = b.process_host(iq.view(np.complex64), shared=True)[0]
        assert r.size > 0 and _bits_equal(a, r), blk


GEOMS = [
    # fs, D, kwargs for both decoders, blocks
    (1.2e6, 5, {}, 8),
    (2.048e6, 9, {}, 8),
    (1.92e6, 8, {}, 8),
    (2.4e6, 11, {"table_size": 100}, 6),       # '%' tuner path (table not a power of two)
    (2.4e6, 11, {"us_version": True}, 6),      # 75 us de-emphasis (FmDecode.cpp:297-298)
    (2.4e6, 11, {"if_filter_order": 250}, 6),  # even order != 8*D
]


@pytest.mark.parametrize("fs,D,kw,nblk", GEOMS)
def test_geometry_bit_exact(oracle, fmsig, fs, D, kw, nblk):
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=11)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, **kw)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, **kw), 1)
    for blk in range(nblk):
        iq = fmsig.generate_f32(p, blk * N, N)
        r = o.process_stream(iq)
        a = b.process_host(iq.view(np.complex64), shared=True)[0]
        assert _bits_equal(a, r), (blk, a.shape, r.shape)
    so, sg = o.status(), b.status()
    assert sg.stereo_detected == so.stereo and sg.rds_state == so.rds_state
    assert np.float32(sg.pilot_level) == np.float32(so.pilot_level)
    assert b.sink.frames.get(0, []) == o.uecp_frames()


@pytest.mark.parametrize("C", [72, 130])
def test_channel_counts_with_padding(oracle, fmsig, C):
    """C = 72 (multiple of 8: XCD-aware mapping, 56 padded lanes) and C = 130 (plain mapping)."""
    pkg = load_package()
    fs, D = 2.4e6, 11
    base = [fmsig.default_params(fs, noise_sigma=0.01, seed=200 + k, f_left=300.0 + 211 * k)
            for k in range(3)]
    check = [0, 1, C // 2, C - 1]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C)
    for blk in range(5):
        src = [fmsig.generate_f32(p, blk * N, N) for p in base]
        iq = np.stack([src[c % 3] for c in range(C)])
        a = b.process_host(iq.view(np.complex64).reshape(C, N))
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(iq[c])), (blk, c)


def test_call_size_limits(fmsig):
    pkg = load_package()
    fs, D = 2.4e6, 11
    d = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    p = fmsig.default_params(fs)
    assert d.ProcessStream(fmsig.generate_f32(p, 0, 65536).view(np.complex64)).size in (2620, 2622)
    assert d.ProcessStream(fmsig.generate_f32(p, 65536, 8192).view(np.complex64)).size > 300
    for bad in (0, 1, 87, 65537):  # 88 = 8 * 11: the smallest call at this geometry
        with pytest.raises(pkg.FmdError):
            d.ProcessStream(np.zeros(bad, np.complex64))
    # downsample = 1: a 65536-sample call would overrun the reference's 32768-entry half-band
    # buffers (DownConvert.cpp:267,500): rejected; half-size calls are fine
    d1 = pkg.FmDecoder(250e3, -37500.0, 48000.0, 15000.0, 1)
    with pytest.raises(pkg.FmdError):
        d1.ProcessStream(np.zeros(65536, np.complex64))
    p1 = fmsig.default_params(250e3)
    assert d1.ProcessStream(fmsig.generate_f32(p1, 0, 32000).view(np.complex64)).size > 10000
    # unsupported configurations fail at construction, loudly
    with pytest.raises(pkg.FmdError):
        pkg.Batch(pkg.make_params(0.0, 0.0), 1)
    # a PCM rate at or below 38 kHz: the reference's 19 kHz notch (FmDecode.cpp:285) is unstable there and its own
    # audio diverges to NaN within a few blocks -- nothing to be bit-exact with
    for pcm in (32000.0, 38000.0):
        with pytest.raises(pkg.FmdError, match="38 kHz"):
            pkg.Batch(pkg.make_params(2.4e6, -0.36e6, pcm, 14400.0, 11), 1)
    pkg.Batch(pkg.make_params(2.4e6, -0.36e6, 38100.0, 15000.0, 11), 1).close()
    # (a baseband rate of 5.33 MHz and more -- the CIC stage -- is decoded since round 6: test_cic_first_stage)
    # the serial stage addresses its row buffers with 32-bit lane offsets: a batch that would pass 4 GB in one of
    # them (9000 channels without decimation: 65552 * 9000 * 8 bytes of IF-FIR output) runs as sub-batches that
    # stay below it (fmd_batch_create: here two of 4544 channels; until round 5 it was refused)
    big = pkg.Batch(pkg.make_params(250e3, -37500.0, downsample=1), 9000, record_callbacks=False)
    assert big.n_channels == 9000 and big.min_samples() > 0
    big.close()


def test_short_calls_bit_exact(oracle, fmsig):
    """Calls far below the usual 65536 samples, decoded like the reference decodes them: half-band
    stages that pass their input on unfiltered (fewer than L inputs, DownConvert.cpp:519-520), stages
    whose delay line is refilled from outputs (fewer than 2 (L - 1) inputs, :546-547), blocks shorter
    than the resampler's history (:236-253), mixed with full blocks so that every history matters.
    2.4 MS/s, D = 11: half-bands of 15, 23 and 43 taps."""
    pkg = load_package()
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=23)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 2)
    nmin = b.min_samples()
    assert nmin == 8 * 11  # 8 baseband samples: 8 -> 4 -> 2 -> 1 through three passing stages
    b.enable_taps()
    start = 0
    sizes = [N, nmin, nmin + 1, 100, 150, 170, 171, 200, 330, 500, 1000, 1900, N, 3000, 3662, 3663, 160, 5000,
             nmin, 97, 8191, 640, 2222, N, 310, 320, 460, 470, 930, 940]
    for k, n in enumerate(sizes):
        iq = fmsig.generate_f32(p, start, n)
        start += n
        ref = o.process_stream(iq)
        a = b.process_host(np.stack([iq, iq]).view(np.complex64))
        t = o.taps()
        for name in ("demod", "baseband", "rds_lpf", "rds_pll", "rds_mf", "mono_rs", "stereo_rs"):
            assert _bits_equal(b.tap(name, 1).view(np.float32), t[name].view(np.float32)), (k, n, name)
        assert _bits_equal(a[0], ref) and _bits_equal(a[1], ref), (k, n)
        so, sg = o.status(), b.status(0)
        for f_o, f_g in ((so.if_level, sg.interface_level), (so.baseband_level, sg.baseband_level),
                         (so.pilot_level, sg.pilot_level)):
            assert np.float32(f_o) == np.float32(f_g), (k, n)
    with pytest.raises(pkg.FmdError):
        b.process_host(np.zeros((2, nmin - 1), np.complex64))
    b.close()


def test_short_calls_long_filter_and_eleven_tap(oracle, fmsig):
    """Blocks shorter than the IF filter (4096 taps: the history keeps part of the previous one,
    DownConvert.cpp:137-145) and short blocks into the 11-tap first stage (needs 20 inputs)."""
    pkg = load_package()
    for fs, D, order, sizes in ((10e6, 46, 4096, [N, 2000, 3000, 4095, 4097, 1000, N, 700, 40000]),
                                (400e3, 1, 0, [32000, 20, 21, 64, 500, 33, 2000, 32000, 25])):
        p = fmsig.default_params(fs, noise_sigma=0.01, seed=29)
        o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order), 1)
        assert min(sizes) >= b.min_samples()
        b.enable_taps()
        start = 0
        for k, n in enumerate(sizes):
            iq = fmsig.generate_f32(p, start, n)
            start += n
            ref = o.process_stream(iq)
            a = b.process_host(iq.view(np.complex64), shared=True)
            t = o.taps()
            for name in ("demod", "rds_lpf", "rds_mf", "mono_rs"):
                assert _bits_equal(b.tap(name).view(np.float32), t[name].view(np.float32)), (fs, k, n, name)
            assert _bits_equal(a[0], ref), (fs, k, n)
        b.close()


@pytest.mark.parametrize("fs,D,sizes", [(400e3, 1, [32000, 32001, 20000, 8193, 32700]),
                                        (644e3, 2, [60000, 65001, 8192, 33333]),
                                        (330e3, 1, [30000, 30001])])
def test_eleven_tap_half_band_first_stage(oracle, fmsig, fs, D, sizes):
    """Baseband rates of 320 kHz and more -- IF rates in [320, 430) kHz and [640, 645) kHz under the
    reference's own downsample rule (RadioReceiver.cpp:285) -- start the RDS decimator with
    CHalfBand11TapDecimateBy2 (DownConvert.cpp:340-341, :589-688): seven products summed as
    written, InLength / 2 outputs, also for odd input lengths.  RDS stage taps and audio bit for bit."""
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=17)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    assert o.rds_hb_lengths()[0] == 11
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 2)
    b.enable_taps()
    start = 0
    for k, n in enumerate(sizes * 2):
        iq = fmsig.generate_f32(p, start, n)
        start += n
        ref = o.process_stream(iq)
        a = b.process_host(np.stack([iq, iq]).view(np.complex64))
        t = o.taps()
        for name in ("rds_lpf", "rds_pll", "rds_mf", "rds_sync"):
            assert _bits_equal(b.tap(name, 1).view(np.float32), t[name].view(np.float32)), (k, n, name)
        assert _bits_equal(a[0], ref) and _bits_equal(a[1], ref), (k, n)
    so, sg = o.status(), b.status(1)
    assert sg.rds_state == so.rds_state and np.float32(sg.pilot_level) == np.float32(so.pilot_level)
    b.close()


@pytest.mark.parametrize("fs,D,ncic,sizes", [(6.4e6, 1, 1, [32000, 20000, 4000, 32000, 2048, 32000]),
                                             (12.0e6, 1, 2, [32000, 16000, 32000, 4096]),
                                             (16.2e6, 3, 1, [65532, 30000, 65532])])
def test_cic_first_stage(oracle, fmsig, fs, D, ncic, sizes):
    """Baseband rates of 5.33 MHz and more start the RDS decimator with CCicN3DecimateBy2 (DownConvert.cpp:340-341,
    :690-727: .125 * (odd + m_Xeven + 3.0 * (m_Xodd + even)), two samples of state), twice from 10.67 MHz up, in front
    of six or seven half-band stages; cRadioReceiver never asks for such a rate (RadioReceiver.cpp:285) but
    cFmDecoder's constructor takes it.  RDS stage taps, audio and getters bit for bit; a call whose baseband length is
    not a multiple of 2 per CIC stage is refused (the class reads one sample past an odd block, :701) and leaves the
    decoder as it was.  The reference's own outputs for two such streams: tests/golden/ref_streams.npz."""
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=27)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    assert o.rds_hb_lengths()[:ncic] == [0] * ncic and o.rds_hb_lengths()[ncic] == 11
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 2)
    b.enable_taps()
    start = 0
    for k, n in enumerate(sizes * 2):
        if k == 3:  # an odd baseband length: refused, nothing changes
            with pytest.raises(pkg.FmdError):
                b.process_host(np.zeros((2, (2 * ncic + 1) * D * 101), np.complex64))
        iq = fmsig.generate_f32(p, start, n)
        start += n
        ref = o.process_stream(iq)
        a = b.process_host(np.stack([iq, iq]).view(np.complex64))
        t = o.taps()
        for name in ("baseband", "mono_rs", "rds_lpf", "rds_pll", "rds_mf", "rds_sync"):
            assert _bits_equal(b.tap(name, 1).view(np.float32), t[name].view(np.float32)), (k, n, name)
        assert _bits_equal(a[0], ref) and _bits_equal(a[1], ref), (k, n)
    so, sg = o.status(), b.status(1)
    assert sg.rds_state == so.rds_state and np.float32(sg.pilot_level) == np.float32(so.pilot_level)
    assert np.float32(sg.interface_level) == np.float32(so.if_level)
    b.close()


@pytest.mark.parametrize("fs,D,order,n", [(1.4e6, 6, 1000, 65536), (1.4e6, 6, 520, 33333),
                                         (2.2e6, 10, 2047, 65536), (10e6, 46, 4096, 40000),
                                         (1.0e6, 4, 600, 65536), (2.4e6, 11, 1000, 65536),
                                         (1.8e6, 8, 1500, 50000), (3.5e6, 16, 1000, 65536),
                                         (2.6e6, 12, 777, 65536)])
def test_long_filters_and_even_decimation(oracle, fmsig, fs, D, order, n):
    """Long IF filters through every window layout of k_if_fir: D = 2*odd (two-region window, the
    hand-scheduled tap loop with its head / tail taps around whole 32-tap pairs of batches, first
    tile out of the history, partial last tile), D = 4*odd (two regions read 16 bytes at a time), D = 8*odd
    and 16*odd (four regions) and odd D (plain, 8 bytes at a time).
    Float and byte input; FIR output and audio bit for bit."""
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=31)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
    o8 = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order), 3)
    b8 = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order), 1)
    b.enable_taps()
    b8.enable_taps()
    for blk in range(3):
        u8 = fmsig.generate_u8(p, blk * n, n)
        iq = oracle.convert_u8(u8)
        ref = o.process_stream(iq)
        a = b.process_host(np.stack([iq, iq, iq]).view(np.complex64))
        for c in (0, 2):
            assert _bits_equal(b.tap("demod", c).view(np.float32), o.taps()["demod"].view(np.float32)), (blk, c)
            assert _bits_equal(a[c], ref), (blk, c)
        ref8 = o8.process_stream_u8(u8)
        a8 = b8.process_host_u8(u8, shared=True)
        assert _bits_equal(b8.tap("demod").view(np.float32), o8.taps()["demod"].view(np.float32)), blk
        assert _bits_equal(a8[0], ref8), blk
    b.close()
    b8.close()
