"""Parity on the dispatch that bench.py times, not only on small batches.

The launch code picks different kernels by batch size: from 1024 channels on the serial stage owns
whole CUs (k_demod_serial<2, true>), with a channel count that is a multiple of 8 the IF FIR uses
the XCD-aware block mapping, overlapped calls (concurrency 2) use two-tile FIR workgroups
(k_if_fir_mt3 / k_if_fir_mt) in the headline geometry and the hand-scheduled tap loops for long filters, and the
post chain runs as a heavy and a light part on two streams.  Small-batch tests never reach those forms,
so these run them at >= 1024 channels with overlapped calls: a handful of channels against the CPU
oracle bit for bit on the very bytes the device generator produced, and the whole batch through a
size-independent property (channels c and c + C/2 carry the same station: identical audio, status
and RDS groups).

Reference being compared: cDownsampleFilter (DownConvert.h:36-39, DownConvert.cpp:98-154) inside
cFmDecoder::ProcessStream (FmDecode.cpp:417-502); ReadAsyncCB (RTL_SDR_Source.cpp:196-213) for
byte input.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _run_overlapped(pkg, fmsig, oracle, fs, D, C, sizes, check, u8, order=0, table=0, lag=2, debug=()):
    """C channels (c and c + C/2 the same station), calls of the given sizes submitted back to back
    in concurrency 2 and consumed `lag` calls late, like bench.py.  Returns nothing; asserts."""
    import torch
    half = C // 2
    chans = [fmsig.channel_params(fs, c % half) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, table_size=table,
                                  if_filter_order=order), C, record_callbacks=False)
    b.set_concurrency(2)
    for key, value in debug:
        b.debug_set(key, value)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    dt = torch.uint8 if u8 else torch.float32
    iq, audio, start = [], [], 0
    for n in sizes:
        n_al = (n + 1) // 2 * 2  # channel stride: a whole number of sample pairs
        t = torch.zeros((C, n_al, 2), dtype=dt, device="cuda")
        if n_al == n:
            gen.generate(t, start, n)
        else:  # ragged length: generate compactly, then spread to the padded stride
            tmp = torch.empty((C, n, 2), dtype=dt, device="cuda")
            gen.generate(tmp, start, n)
            t[:, :n] = tmp
        iq.append((t, n, n_al))
        audio.append(torch.zeros((C, a_stride), dtype=torch.float32, device="cuda"))
        start += n
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    nf, groups = [], []
    for k, (t, n, n_al) in enumerate(iq):
        nf.append(b.process_device(t.data_ptr(), n_al, n, audio[k].data_ptr(), a_stride, st, u8=u8))
        if k >= lag:
            b.wait(stream=st, lag=lag)
            groups.append(b.collect_rds_array(cap=4 * C, stream=st, lag=lag))
    b.wait(stream=st)
    groups.append(b.collect_rds_array(cap=4 * C, stream=st))
    torch.cuda.synchronize()

    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, table_size=table,
                                    if_filter_order=order) for c in check}
    for k, (t, n, n_al) in enumerate(iq):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for c in check:
            x = t[c, :n].cpu().numpy().reshape(-1)
            r = refs[c].process_stream_u8(x) if u8 else refs[c].process_stream(x)
            assert _bits_equal(a[c], r), (k, c, n)
        assert np.array_equal(a[:half].view(np.uint32), a[half:].view(np.uint32)), (k, n)
    g = np.concatenate(groups)
    lo, hi = g[g["channel"] < half], g[g["channel"] >= half]
    key = lambda x, off: sorted((int(c) - off, int(k), tuple(int(v) for v in bl))
                                for c, k, bl in zip(x["channel"], x["call_index"], x["blocks"]))
    assert key(lo, 0) == key(hi, half)
    for c in check:
        mine = sorted((int(k), tuple(int(v) for v in bl)) for ch, k, bl in
                      zip(g["channel"], g["call_index"], g["blocks"]) if ch == c)
        assert mine == sorted((k, tuple(bl)) for k, bl in refs[c].rds_groups()), c
        so, sg = refs[c].status(), b.status(c)
        assert sg.stereo_detected == so.stereo
        for f_o, f_g in ((so.if_level, sg.interface_level), (so.baseband_level, sg.baseband_level),
                         (so.pilot_level, sg.pilot_level), (so.tuning_offset, sg.tuning_offset)):
            assert np.float32(f_o) == np.float32(f_g), c
    b.close()


@pytest.mark.parametrize("u8", [False, True], ids=["f32", "u8"])
def test_config5_on_its_benchmarked_dispatch(oracle, fmsig, u8):
    """BASELINE config 5 (4096-tap cDownsampleFilter, 10 MS/s, D = 46) the way `bench.py --workload
    config5` runs it: channel count a multiple of 8 and >= 1024 (XCD-aware mapping, TILE 256,
    hand-scheduled tap loop, whole-CU serial stage), calls overlapped, outputs consumed late."""
    pkg = load_package()
    C = 1032
    _run_overlapped(pkg, fmsig, oracle, 10e6, 46, C, [N] * 4,
                    check=[0, 1, 7, 8, 515, 516, 1030, 1031], u8=u8, order=4096)


@pytest.mark.parametrize("fs,D,order", [(2.4e6, 11, 2048), (9.6e6, 44, 2048), (1.0e6, 4, 1500),
                                        (1.4e6, 6, 1024)],
                         ids=["odd-D", "4x-odd-D", "D4", "2x-odd-D"])
def test_long_filter_tap_loops_overlapped(oracle, fmsig, fs, D, order):
    """The hand-scheduled long-filter tap loops of every window layout -- plain window read 8 bytes at a
    time (odd D: fir_long_odd_asm), four regions (D = 4 * odd, D = 4: fir_long_e2_asm), plain window read 16
    bytes at a time (D = 2 * odd: fir_long_b128_asm) -- at >= 1024 channels (TILE 256, XCD-aware block
    mapping, whole-CU serial stage) with overlapped calls and late consumption, on full and ragged
    blocks (head / tail taps around the 32-tap pairs of batches, first tile out of the history)."""
    pkg = load_package()
    _run_overlapped(pkg, fmsig, oracle, fs, D, 1032, [N, 40001, N, 12288],
                    check=[0, 1, 8, 515, 1024, 1031], u8=False, order=order)


@pytest.mark.parametrize("ro", [3, 2, 1], ids=["3-per-lane", "2-per-lane", "1-per-lane"])
@pytest.mark.parametrize("u8", [False, True], ids=["f32", "u8"])
def test_config4_geometry_overlapped_ragged_two_tile_fir(oracle, fmsig, u8, ro):
    """The headline geometry (2.4 MS/s, D = 11, 88 taps) at >= 1024 channels with overlapped calls: the
    two-tile IF FIR forms -- k_if_fir_mt3 with three and two (what ships) outputs per lane, k_if_fir_mt
    with one -- and the whole-CU serial stage, on ragged call sizes: odd lengths, a partial last tile as
    the second tile of a workgroup, an odd tile count (the early `tile >= ntiles` exit), lanes whose last
    outputs lie beyond the call, a short first-tile history -- and on full blocks."""
    pkg = load_package()
    C = 1024
    # outputs per call ~ n / 11, tiles of 64 x ro outputs, two tiles per workgroup
    sizes = [N, 10007, 8192, 65535, 150, 33001, 1001, 45057, N, 8193, 330, 21120]  # incl. short calls
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, C, sizes,
                    check=[0, 63, 64, 511, 512, 1023], u8=u8, debug=(("fir_ro", ro),))


def test_config4_geometry_channel_count_not_a_multiple_of_8(oracle, fmsig):
    """The same dispatch with 1030 channels: no XCD-aware block map (blocks channel-major), a ragged last
    group of 6 channels in every lane-per-channel kernel."""
    pkg = load_package()
    C = 1030
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, C, [N, 10007, 2113, 65535, 33001, N],
                    check=[0, 1, 514, 515, 1028, 1029], u8=False)


def test_config3_shared_capture_overlapped(oracle, fmsig):
    """BASELINE config 3 (256 channels from ONE capture, table_size 256) with overlapped calls and
    late consumption, as `bench.py --workload config3` runs it."""
    import torch
    pkg = load_package()
    fs, D, C, T, nblk, lag = 2.4e6, 11, 256, 256, 6, 2
    stations = [fmsig.default_params(fs, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i,
                                     pi=0x5000 + i, ps="CAP%05d" % i, f_left=500.0 + 300 * i)
                for i, f0 in enumerate((-600e3, -360e3, -150e3, 75e3, 300e3, 600e3))]
    shifts = np.arange(C, dtype=np.int32) - 128
    b = pkg.Batch(pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T), C,
                  tuning_shifts=shifts, record_callbacks=False)
    b.set_concurrency(2)
    check = [0, 64, 90, 112, 136, 160, 192, 255]
    refs = {c: oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T,
                                    tuning_shift=int(shifts[c])) for c in check}
    caps = []
    for blk in range(nblk):
        cap = np.zeros(2 * N, dtype=np.float32)
        for p in stations:
            cap += fmsig.generate_f32(p, blk * N, N)
        caps.append(cap)
    d_cap = [torch.from_numpy(c).cuda() for c in caps]
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    st = torch.cuda.current_stream().cuda_stream
    nf = []
    for k in range(nblk):
        nf.append(b.process_device(d_cap[k].data_ptr(), 0, N, audio[k].data_ptr(), a_stride, st))
        if k >= lag:
            b.wait(stream=st, lag=lag)
    b.wait(stream=st)
    torch.cuda.synchronize()
    for k in range(nblk):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(caps[k])), (k, c)
    b.close()


def test_process_host_is_synchronous_in_concurrency_2(oracle, fmsig):
    """fmd_batch_process_host returns finished audio in every concurrency mode: in mode 2 the null
    stream is not ordered behind the call, so the entry point has to order it behind the call's last
    kernel (the audio tail on the light stream) before it copies the audio out."""
    pkg = load_package()
    fs, D, C = 2.4e6, 11, 3
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=40 + c, pi=0x4400 + c) for c in range(C)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C)
    b.set_concurrency(2)
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(C)]
    for blk in range(6):
        iq = np.stack([fmsig.generate_f32(ps[c], blk * N, N) for c in range(C)])
        audio = b.process_host(iq.view(np.complex64).reshape(C, N))
        for c in range(C):
            assert _bits_equal(audio[c], refs[c].process_stream(iq[c])), (blk, c)
    b.close()


def test_profiling_level_change_between_overlapped_calls(oracle, fmsig):
    """fmd_batch_set_profiling(2) moves the next call onto the caller's stream with no event waits;
    made between overlapped calls it must first let the calls in flight finish."""
    import torch
    pkg = load_package()
    fs, D, C = 2.4e6, 11, 64
    chans = [fmsig.channel_params(fs, c) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    nblk = 8
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    for k in range(nblk):
        gen.generate(iq[k], k * N, N)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    nf = []
    for k in range(nblk):
        if k == 3:
            b.set_profiling(2)  # no drain by the caller
        if k == 6:
            b.set_profiling(0)
        nf.append(b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), a_stride, st))
    b.wait(stream=st)
    torch.cuda.synchronize()
    check = [0, 31, 63]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    for k in range(nblk):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(iq[k][c].cpu().numpy().reshape(-1))), (k, c)
    b.close()


def test_device_error_word_surfaces(fmsig):
    """A serial-stage hand-off that times out does not go unnoticed: the kernel sets a bit in the
    batch's error word, fmd_batch_wait / collect_rds return FMD_ERR_DEVICE, later calls are refused
    until fmd_batch_reset.  fmd_batch_debug_set_spin_limit(b, 0) makes every hand-off wait time out at
    once."""
    import torch
    pkg = load_package()
    fs, D, C = 2.4e6, 11, 1024  # >= 1024 channels: the whole-CU serial stage with LDS hand-offs
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.debug_set_spin_limit(0)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = torch.zeros((C, N, 2), dtype=torch.float32, device="cuda")
    audio = torch.zeros((C, a_stride), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    b.process_device(iq.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
    torch.cuda.synchronize()
    with pytest.raises(pkg.FmdError, match="hand-off timed out"):
        b.wait(stream=st)
        torch.cuda.synchronize()
        b.collect_rds_array(cap=16, stream=st)
    with pytest.raises(pkg.FmdError, match="hand-off timed out"):
        b.process_device(iq.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
    b.reset()  # clears the failure: the batch takes calls again
    b.close()

    # the same batch size without the knob: no error
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    b.process_device(iq.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
    b.wait(stream=st)
    torch.cuda.synchronize()
    b.collect_rds_array(cap=16, stream=st)
    b.close()


def test_device_timeline_of_overlapped_calls(fmsig):
    """fmd_batch_debug_timeline (what `FMD_BENCH_TIMELINE=1 python bench.py` prints): per profiled call the
    device times of its IF FIR, serial stage and audio tail.  Every call's stages are in order, the
    serial stages of consecutive calls do not overlap (they hand channel state to each other), and
    the FIR of call k + 1 does overlap the serial stage of call k -- the point of concurrency 2."""
    import torch
    pkg = load_package()
    fs, D, C, nblk = 2.4e6, 11, 4096, 8  # (4096 channels: the large-batch forms of half-band chain and resampler)
    gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c % 16) for c in range(C)], "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    for k in range(nblk):
        gen.generate(iq[k], k * N, N)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    b.set_profiling(1)
    for k in range(nblk):
        b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), a_stride, st)
    b.wait(stream=st)
    torch.cuda.synchronize()
    tl = b.debug_timeline()
    assert tl.shape == (nblk, 10) and (tl >= 0).all() and tl[0, 0] == 0.0
    for k in range(nblk):
        f0, f1, s0, s1, t0, t1, h0, h1, r0, r1 = tl[k]
        assert f0 < f1 <= s0 < s1 <= t0 < t1, (k, tl[k])
        assert s1 <= h0 < h1 <= r0 < r1 <= t0, (k, tl[k])  # half-band chain, then resampler, behind the serial stage
    assert all(tl[k, 3] <= tl[k + 1, 2] for k in range(nblk - 1))           # serial stages one after the other
    assert any(tl[k + 1, 0] < tl[k, 3] for k in range(1, nblk - 1))           # a later FIR beside an earlier serial stage
    b.set_profiling(0)
    b.close()


@pytest.mark.parametrize("debug", [(("split_post", 1),), (("split_post", 1), ("halfband_chain", 1), ("lpf_late", 0))],
                         ids=["split_post", "split_post+chain"])
def test_two_post_streams_overlapped_at_4096_channels(oracle, fmsig, debug):
    """The post chain on two streams (what batches of more than 16384 channels take, here forced at 4096)
    with overlapped calls: without mixed rows the half-band chain's first stage reads history rows of the
    baseband buffer that the PREVIOUS call's roll wrote on the other stream -- the chain has to wait for
    that call's EV_ROLL.  Audio, status and RDS groups bit for bit, eight back-to-back calls."""
    pkg = load_package()
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, 4096, [N, N, 30001, N, N, 12345, N, N],
                    check=[0, 63, 2047, 2048, 4095], u8=False, debug=debug)


@pytest.mark.parametrize("layout", [0, 1, 2], ids=["heavy-stream", "own-stream", "light-streams"])
def test_stream_layouts_of_an_overlapped_call(oracle, fmsig, layout):
    """The three stream layouts of an overlapped call (fmd_batch_debug_set "lpf_late"; process_device_impl): the
    post chain's two complex low-pass filters (cFirFilter::Process / ProcessTwo, FirFilter.cpp:330-413) on the heavy
    stream (what long IF filters take), on a stream of their own (round 4's), or at the heads of the light part's
    two streams (the default for short IF filters) -- on ragged calls: fewer audio frames than taps, full blocks,
    consumed two calls late.  Same audio, status and groups bit for bit."""
    pkg = load_package()
    sizes = [N, 10007, 1001, 330, 65535, 150, 33001, N, 2000, 21120, N]
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, 1024, sizes, check=[0, 63, 64, 511, 1023], u8=False,
                    debug=(("lpf_late", layout),))


def test_stream_layout_may_change_between_calls(oracle, fmsig):
    """The layouts keep the filters' delay lines in the same rows (the front rows of the decimator's and the
    resampler's output buffers, rolled by whoever ran the filter): a batch may change layout between any two
    calls (the key drains the device first)."""
    pkg = load_package()
    import torch
    fs, D, C = 2.4e6, 11, 1024
    chans = [fmsig.channel_params(fs, c % 4) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    check = [0, 5, 1023]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for k, n in enumerate([N, 30002, 500, N, 12346, 200, N, N, N]):
        b.debug_set("lpf_late", k % 3)
        t = torch.zeros((C, n, 2), dtype=torch.float32, device="cuda")
        gen.generate(t, pos, n)
        audio = torch.zeros((C, a_stride), dtype=torch.float32, device="cuda")
        nf = b.process_device(t.data_ptr(), n, n, audio.data_ptr(), a_stride, st)
        b.wait(stream=st)
        torch.cuda.synchronize()
        a = audio[:, :nf].cpu().numpy()
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(t[c].cpu().numpy().reshape(-1))), (k, c, n)
        pos += n
    b.close()


@pytest.mark.parametrize("u8", [False, True], ids=["f32", "u8"])
def test_batch_above_8192_channels_runs_as_sub_batches(oracle, fmsig, u8):
    """One fmd_batch of 16 448 channels = a shell over three sub-batches (5504 + 5504 + 5440) that share the five
    internal streams; a call is submitted sub-batch by sub-batch into one sequence of whole-CU pipeline calls
    (fmd_batch::subs, csrc/fmd_batch.hip).  Channels are independent (FmDecode.h:201-212): the first and last
    channel of every sub-batch against the CPU oracle bit for bit, and c / c + C/2 (which lie in DIFFERENT sub-batches)
    identical in audio, status and RDS groups -- full and ragged calls, consumed two calls late."""
    pkg = load_package()
    C = 16448
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, C, [N, 30001, N, N],
                    check=[0, 5503, 5504, 8223, 8224, 11007, 11008, 16447], u8=u8)


def test_sub_batches_through_every_entry_point(oracle, fmsig):
    """The rest of the batch API over a shell (8320 channels = 4224 + 4096): the host-buffer entry point and the
    device one in the default concurrency mode (caller's stream ordered after every call), Reset, the getters, a
    stage tap, group decoder callbacks with the caller's channel numbers, device export with a channel offset."""
    import torch
    pkg = load_package()
    fs, D, C = 2.4e6, 11, 8320
    chans = [fmsig.channel_params(fs, c % 4) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=True)
    check = [0, 4223, 4224, 8319]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    pos = 0
    # host buffers in and out, short calls, a Reset in the middle (FmDecode.cpp:326-338)
    for k, n in enumerate([16384, 8192, 16384, 12000, 16384]):
        t = torch.empty((C, n, 2), dtype=torch.float32, device="cuda")
        gen.generate(t, pos, n)
        iq = t.cpu().numpy().reshape(C, 2 * n).view(np.complex64)
        a = b.process_host(iq)
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(iq[c])), (k, c)
        assert np.array_equal(a[:4].view(np.uint32), a[C - 4:].view(np.uint32))  # same stations (C % 4 == 0)
        if k == 2:
            b.reset()
            for c in check:
                refs[c].reset()
        pos += n
    # device buffers, full blocks; the groups through the shell's group decoders (callbacks), then by device export
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    t = torch.empty((C, N, 2), dtype=torch.float32, device="cuda")
    audio = torch.zeros((C, a_stride), dtype=torch.float32, device="cuda")
    rec = torch.zeros((4 * C, 4), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    exported = {c: [] for c in check}
    for k in range(22):
        gen.generate(t, pos, N)
        torch.cuda.synchronize()
        nf = b.process_device(t.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
        torch.cuda.synchronize()  # (mode 1: the caller's stream is behind the call)
        a = audio[:, :nf].cpu().numpy()
        for c in check:
            assert _bits_equal(a[c], refs[c].process_stream(t[c].cpu().numpy().reshape(-1))), (k, c)
        if k < 14:
            b.collect_rds_array(cap=4 * C, run_group_decoder=True, stream=st)
            if k == 13:
                for c in check:
                    assert b.sink.frames.get(c, []) == refs[c].uecp_frames(), c
                assert b.sink.frames.get(4, []) == b.sink.frames.get(0, [])
        else:
            b.export_rds_device(rec.data_ptr(), 4 * C, channel_offset=1000, stream=st)
            torch.cuda.synchronize()
            r = rec.cpu().numpy()
            r = r[r[:, 0] != 0]
            ch = r[:, 0] - 1001
            assert ((ch >= 0) & (ch < C)).all()
            for c in check:
                exported[c] += [(int(row[1]), (int(row[2]) & 0xffff, (int(row[2]) >> 16) & 0xffff,
                                               int(row[3]) & 0xffff, (int(row[3]) >> 16) & 0xffff)) for row in r[ch == c]]
            counts = np.bincount(ch, minlength=C)
            assert (counts.reshape(-1, 4) == counts[:4]).all()  # every channel of a station reports as many groups
        pos += N
    for c in check:
        first_exported = min((k for k, _ in exported[c]), default=None)
        assert first_exported is not None
        want = [(k, tuple(bl)) for k, bl in refs[c].rds_groups() if k >= first_exported]
        assert sorted(exported[c]) == sorted(want), c
        so, sg = refs[c].status(), b.status(c)
        assert sg.stereo_detected == so.stereo
        for f_o, f_g in ((so.if_level, sg.interface_level), (so.baseband_level, sg.baseband_level),
                         (so.pilot_level, sg.pilot_level), (so.tuning_offset, sg.tuning_offset)):
            assert np.float32(f_o) == np.float32(f_g), c
        assert _bits_equal(b.tap("mono_rs", c), b.tap("mono_rs", c % 4))
    b.close()


@pytest.mark.parametrize("excl", [0, 1], ids=["shared-form", "whole-cu-form"])
def test_serial_stage_forms_at_the_other_batch_size(oracle, fmsig, excl):
    """"serial_exclusive" of fmd_batch_debug_set: the serial stage's whole-CU form (one role wave per SIMD of a CU:
    the default for 1024-8192 channels) forced on a batch that would take the shared form (512 channels) and the
    other way round (1024) -- FM PLL, pilot PLL and level meters (FmDecode.cpp:362-415, 143-229, 522-539) bit for
    bit either way, overlapped calls, ragged sizes."""
    pkg = load_package()
    C = 512 if excl else 1024
    _run_overlapped(pkg, fmsig, oracle, 2.4e6, 11, C, [N, 30001, N, 8192, N], check=[0, 63, 64, C // 2 - 1, C - 1],
                    u8=False, debug=(("serial_exclusive", excl),))


def test_stage_mask_leaves_the_launched_part_intact(oracle, fmsig):
    """"stage_mask" (tools/power_by_stage.py's joules per part: only the named parts of a call are launched, every
    event is still recorded): with only the IF stage launched the calls still complete in order and the IF FIR's
    output (cFineTuner + cDownsampleFilter, DownConvert.cpp:98-154) is what the full path's IF stage writes."""
    import torch
    pkg = load_package()
    fs, D, C = 2.4e6, 11, 1024
    gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c % 8) for c in range(C)], "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    b.debug_set("stage_mask", 1)
    ref = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    audio = torch.zeros((C, a_stride), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for k in range(4):
        t = torch.empty((C, N, 2), dtype=torch.float32, device="cuda")
        gen.generate(t, k * N, N)
        b.process_device(t.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
        b.wait(stream=st)
        torch.cuda.synchronize()
        ref.process_stream(t[5].cpu().numpy().reshape(-1))
        assert _bits_equal(b.tap("demod", 5).view(np.float32), ref.taps()["demod"].view(np.float32)), k
    b.debug_set("stage_mask", 63)
    b.close()
