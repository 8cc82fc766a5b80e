"""RTL-SDR byte ingest fused into the IF kernel.

The reference converts every librtlsdr byte with float(b / (255.0 / 2.0) - 1.0) in
cRtlSdrSource::ReadAsyncCB (RTL_SDR_Source.cpp:207-211) and hands the complex<float> block to
ProcessStream.  The u8 entry points do the conversion inside the FIR kernel; their output has to
equal the oracle's ReadAsyncCB restatement followed by its ProcessStream, bit for bit.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

AUDIO_RMS_TOL = 1e-5  # BASELINE.json: "float audio within 1e-5 RMS"
N = 65536


def _bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2))) if a.size else 0.0


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_all_byte_values_convert_like_the_reference(pkg, oracle):
    """A block that holds every (I, Q) byte pair: the demodulator input tap (tuned, filtered,
    decimated) only equals the oracle's if each of the 256 byte values converts exactly."""
    fs, D = 2.4e6, 11
    rng = np.random.default_rng(5)
    buf = np.empty(2 * N, dtype=np.uint8)
    buf[0::2] = np.tile(np.arange(256, dtype=np.uint8), N // 256)
    buf[1::2] = np.repeat(np.arange(256, dtype=np.uint8), N // 256)
    buf = buf.reshape(-1, 2)[rng.permutation(N)].reshape(-1)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    b.enable_taps()
    for _ in range(2):
        o.process_stream_u8(buf)
        b.process_host_u8(buf, shared=True)
        assert _bits_equal(b.tap("demod").view(np.float32), o.taps()["demod"].view(np.float32))
    b.close()


@pytest.mark.parametrize("fs,D", [(2.4e6, 11), (1.0e6, 4)])
def test_single_decoder_u8_equals_convert_then_process(pkg, oracle, fmsig, fs, D):
    """cFmDecoder surface: ProcessStreamU8(bytes) == oracle ReadAsyncCB + ProcessStream, audio
    bit for bit, UECP frames and PS name identical; and equal to the product's own float path."""
    p = fmsig.default_params(fs, noise_sigma=0.005)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    d8 = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    df = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    worst = 0.0
    for blk in range(40):
        buf = fmsig.generate_u8(p, blk * N, N)
        a_ref = o.process_stream_u8(buf)
        a_u8 = d8.ProcessStreamU8(buf)
        a_f = df.ProcessStream(oracle.convert_u8(buf).view(np.complex64))
        assert a_u8.shape == a_ref.shape
        worst = max(worst, _rms(a_u8, a_ref))
        assert _bits_equal(a_u8, a_ref), "block %d" % blk
        assert _bits_equal(a_u8, a_f)
    assert worst <= AUDIO_RMS_TOL
    assert d8.sink.frames.get(0, []) == o.uecp_frames()
    assert d8.sink.names.get(0) == o.channel_name()
    assert len(o.rds_groups()) > 10


@pytest.mark.parametrize("n", [65536, 20001, 8192, 33333])
def test_batch_u8_ragged_blocks_and_shifts(pkg, oracle, fmsig, n):
    """Several channels with their own tuner shifts, block lengths that are not multiples of the
    16-byte load (the kernel's sample-by-sample tail), stage taps and audio bit for bit."""
    fs, D = 2.4e6, 11
    shifts = [10, -7, 0, 31, 10]
    Cn = len(shifts)
    ps = [fmsig.channel_params(fs, c) for c in range(Cn)]
    os_ = [oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, tuning_shift=s) for s in shifts]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), Cn, tuning_shifts=shifts)
    b.enable_taps()
    for blk in range(6):
        bufs = np.stack([fmsig.generate_u8(ps[c], blk * n, n) for c in range(Cn)])
        a = b.process_host_u8(bufs)
        for c in range(Cn):
            a_ref = os_[c].process_stream_u8(bufs[c])
            assert _bits_equal(b.tap("demod", c).view(np.float32),
                               os_[c].taps()["demod"].view(np.float32)), (blk, c)
            assert _bits_equal(a[c], a_ref), (blk, c)
    for c in range(Cn):
        so, sg = os_[c].status(), b.status(c)
        assert np.float32(so.if_level) == np.float32(sg.interface_level)
    b.close()


def test_float_path_ragged_blocks(pkg, oracle, fmsig):
    """Same ragged block lengths through the complex<float> entry point (odd lengths exercise the
    tail of its two-sample loads)."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01)
    for n in (20001, 8193):
        o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
        for blk in range(5):
            iq = fmsig.generate_f32(p, blk * n, n)
            assert _bits_equal(b.process_host(iq.view(np.complex64), shared=True)[0],
                               o.process_stream(iq)), (n, blk)
        b.close()


def test_device_entry_u8_rejects_misaligned_stride(pkg):
    import torch
    fs, D = 2.4e6, 11
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 2)
    iq = torch.zeros(2 * 2 * N + 64, dtype=torch.uint8, device="cuda")
    audio = torch.zeros(2 * b.max_audio_floats(N), dtype=torch.float32, device="cuda")
    with pytest.raises(pkg.FmdError):
        b.process_device(iq.data_ptr(), N + 1, N, audio.data_ptr(), b.max_audio_floats(N), u8=True)
    with pytest.raises(pkg.FmdError):
        b.process_device(iq.data_ptr() + 2, N, N, audio.data_ptr(), b.max_audio_floats(N), u8=True)
    b.close()


def test_u8_batch_overlapped_whole_cu_form(pkg, oracle, fmsig):
    """Byte input through a 1030-channel batch with overlapped calls: the byte instantiation of the
    two-tile FIR workgroups beside the whole-CU serial stage.  Eight distinct stations repeated over
    the channels: one channel of each against ReadAsyncCB + ProcessStream of the oracle, all the
    others against their twin."""
    import torch
    fs, D, C, nblk, K = 2.4e6, 11, 1030, 4, 8
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=31 + s) for s in range(K)]
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(K)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    st = torch.cuda.current_stream().cuda_stream
    stride = (b.max_audio_floats(N) + 63) // 64 * 64
    idx = torch.arange(C, device="cuda") % K
    base, bufs, outs, nfs = [], [], [], []
    for k in range(nblk):
        u8 = np.stack([fmsig.generate_u8(ps[s], k * N, N) for s in range(K)])  # [K, 2N] bytes
        base.append(u8)
        d_iq = torch.from_numpy(u8).cuda()[idx].contiguous()
        d_out = torch.zeros((C, stride), dtype=torch.float32, device="cuda")
        bufs.append(d_iq)
        outs.append(d_out)
        nfs.append(b.process_device(d_iq.data_ptr(), N, N, d_out.data_ptr(), stride, st, u8=True))
    b.wait(stream=st)
    torch.cuda.synchronize()
    for k in range(nblk):
        a = outs[k][:, :nfs[k]].cpu().numpy()
        for s in range(K):
            r = refs[s].process_stream(oracle.convert_u8(base[k][s]))
            assert _bits_equal(a[s], r), (k, s)
            twins = a[s::K]
            assert np.array_equal(twins.view(np.uint32), np.broadcast_to(a[s], twins.shape).view(np.uint32)), (k, s)
    b.close()
