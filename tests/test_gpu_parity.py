"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, stage by stage and
end to end, on the same seeded synthetic IQ.

Bars (BASELINE.json north_star): float audio within 1e-5 RMS of the CPU path, RDS group
bits bit-exact.  The kernels are written to be bit-faithful, so the stage taps are compared
for exact equality first and the RMS gate is reported as the contract.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

AUDIO_RMS_TOL = 1e-5  # BASELINE.json: "float audio within 1e-5 RMS"
N = 65536


def _rms(a, b):
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2))) if a.size else 0.0


def _bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_design_matches_oracle(pkg, oracle):
    """Host-side constants/taps of the product equal the oracle's restatement bit for bit."""
    for fs, D in ((2.4e6, 11), (1.0e6, 4), (10e6, 46)):
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
        o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
        assert _bits_equal(b.design("if_taps"), o.if_taps())
        assert _bits_equal(b.design("audio_lpf"), o.audio_taps())
        assert _bits_equal(b.design("rds_lpf"), o.rds_lpf_taps())
        assert _bits_equal(b.design("rds_mf"), o.rds_mf_taps())
        assert _bits_equal(b.design("lut0"), o.lut().view(np.float32))
        sc, oc = b.scalars(), o.constants()
        for k, v in oc.items():
            assert np.float32(sc[k]) == np.float32(v), (fs, k, sc[k], v)
        b.close()


STAGES = ["demod", "baseband", "pilot38", "mono_rs", "stereo_rs", "rds_lpf", "rds_pll", "rds_mf",
          "rds_sync"]


@pytest.mark.parametrize("fs,D,nblk", [(2.4e6, 11, 20), (1.0e6, 4, 10)])
def test_stage_taps_bit_exact(pkg, oracle, fmsig, fs, D, nblk):
    """Every stage output of every block equals the oracle's, bit for bit (noisy stereo+RDS)."""
    p = fmsig.default_params(fs, noise_sigma=0.01)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    b.enable_taps()
    for blk in range(nblk):
        iq = fmsig.generate_f32(p, blk * N, N)
        a_ref = o.process_stream(iq)
        a_gpu = b.process_host(iq.view(np.complex64), shared=True)[0]
        taps = o.taps()
        for name in STAGES:
            g = b.tap(name)
            r = taps[name]
            assert g.shape == r.shape, (blk, name, g.shape, r.shape)
            assert _bits_equal(g.view(np.float32), r.view(np.float32)), \
                "block %d stage %s: rms diff %g" % (blk, name, _rms(g, r))
        assert a_gpu.shape == a_ref.shape
        assert _rms(a_gpu, a_ref) <= AUDIO_RMS_TOL
        assert _bits_equal(a_gpu, a_ref), "block %d audio: rms diff %g" % (blk, _rms(a_gpu, a_ref))
        so, sg = o.status(), b.status()
        assert sg.stereo_detected == so.stereo
        for f_o, f_g in ((so.if_level, sg.interface_level), (so.baseband_level, sg.baseband_level),
                         (so.pilot_level, sg.pilot_level), (so.tuning_offset, sg.tuning_offset)):
            assert np.float32(f_o) == np.float32(f_g)
    b.close()


def test_config2_stereo_rds_six_seconds(pkg, oracle, fmsig):
    """BASELINE config 2: 1 stereo channel + RDS, 2.4 MS/s, >= 6 s through the cFmDecoder
    surface: audio <= 1e-5 RMS on every block, every RDS group identical and in the same
    call, UECP frames identical, PS name delivered."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.005)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    d = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    nblk = int(6.0 * fs / N) + 1
    worst = 0.0
    for blk in range(nblk):
        iq = fmsig.generate_f32(p, blk * N, N)
        a_ref = o.process_stream(iq)
        a_gpu = d.ProcessStream(iq.view(np.complex64))
        assert a_gpu.shape == a_ref.shape
        worst = max(worst, _rms(a_gpu, a_ref))
    assert worst <= AUDIO_RMS_TOL
    assert d.StereoDetected() and o.status().stereo == 1
    assert d.sink.frames.get(0, []) == o.uecp_frames()
    assert d.sink.names.get(0) == o.channel_name() == "TESTFM01"
    assert len(o.rds_groups()) > 50


def test_rds_groups_bit_exact_and_call_aligned(pkg, oracle, fmsig):
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.02, pi=0xC0DE, ps="GPU RDS ")
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    got = []
    for blk in range(80):
        iq = fmsig.generate_f32(p, blk * N, N)
        o.process_stream(iq)
        b.process_host(iq.view(np.complex64), shared=True)
    # process_host already drained the queue through the group decoder; compare its frames
    assert b.sink.frames.get(0, []) == o.uecp_frames()
    # raw group path: run again from a fresh pair and collect groups explicitly
    o2 = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b2 = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    import torch
    a_dev = torch.zeros(b2.max_audio_floats(N), dtype=torch.float32, device="cuda")
    for blk in range(80):
        iq = fmsig.generate_f32(p, blk * N, N)
        o2.process_stream(iq)
        x = torch.from_numpy(iq).cuda()
        b2.process_device(x.data_ptr(), 0, N, a_dev.data_ptr(), a_dev.numel(),
                          torch.cuda.current_stream().cuda_stream)
        got += [(ci, blocks) for _ch, ci, blocks in
                b2.collect_rds(stream=torch.cuda.current_stream().cuda_stream)]
    assert got == o2.rds_groups()
    assert len(got) > 10


@pytest.mark.parametrize("fs,D,a_rds,noise,seed", [(2.4e6, 11, 0.012, 0.05, 5), (2.4e6, 11, 0.011, 0.04, 10),
                                                   (1.0e6, 4, 0.012, 0.05, 6)])
def test_weak_rds_sync_loss_and_error_correction(pkg, oracle, fmsig, fs, D, a_rds, noise, seed):
    """RDS subcarrier near the decoding threshold: blocks arrive with bit errors, the syndrome
    decoder corrects bursts (CheckBlock, RDSProcess.cpp:363-431), block sync is lost and found
    again (ProcessNewRdsBit states, :272-361).  The group stream, the state after every call and
    the audio must still be the oracle's, bit for bit."""
    p = fmsig.default_params(fs, noise_sigma=noise, a_rds=a_rds, seed=seed)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1, record_callbacks=False)
    import torch
    st = torch.cuda.current_stream().cuda_stream
    a_dev = torch.zeros(b.max_audio_floats(N), dtype=torch.float32, device="cuda")
    got, states_o, states_g = [], [], []
    for blk in range(150):
        iq = fmsig.generate_f32(p, blk * N, N)
        a_ref = o.process_stream(iq)
        x = torch.from_numpy(iq).cuda()
        nf = b.process_device(x.data_ptr(), 0, N, a_dev.data_ptr(), a_dev.numel(), st)
        got += [(ci, blocks) for _ch, ci, blocks in b.collect_rds(stream=st)]
        assert _bits_equal(a_dev[:nf].cpu().numpy(), a_ref), blk
        states_o.append(o.status().rds_state)
        states_g.append(b.status().rds_state)
    assert states_g == states_o
    assert len(set(states_o)) >= 3  # bit sync, block sync and group decode were all visited
    assert sum(1 for i in range(1, 150) if states_o[i] < states_o[i - 1]) >= 2  # sync was lost
    assert got == o.rds_groups() and len(got) >= 10
    b.close()


def test_batch_channels_independent(pkg, oracle, fmsig):
    """Several different stations in one batch: each channel equals its own oracle run."""
    fs, D = 2.4e6, 11
    C = 5
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=100 + c, f_left=400.0 + 150 * c,
                               f_right=3000.0 - 170 * c, pi=0x1000 + c, ps="CHAN%04d" % c)
          for c in range(C)]
    os_ = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(C)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C)
    for blk in range(12):
        iq = np.stack([fmsig.generate_f32(ps[c], blk * N, N) for c in range(C)])
        a = b.process_host(iq.view(np.complex64).reshape(C, N))
        for c in range(C):
            r = os_[c].process_stream(iq[c])
            assert _bits_equal(a[c], r), (blk, c, _rms(a[c], r))
    for c in range(C):
        assert b.sink.frames.get(c, []) == os_[c].uecp_frames()


def test_mono_station_config1_geometry(pkg, oracle, fmsig):
    """config-1 input (mono, noise 0.01) on the GPU path: stays mono, L == R, equals the oracle."""
    fs, D = 2.4e6, 11
    p = fmsig.mono_params(fs, noise_sigma=0.01)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    d = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    for blk in range(24):
        iq = fmsig.generate_f32(p, blk * N, N)
        r = o.process_stream(iq)
        a = d.ProcessStream(iq.view(np.complex64))
        assert _bits_equal(a, r)
    assert not d.StereoDetected()
    assert np.array_equal(a[0::2], a[1::2])


def test_ragged_block_sizes_and_reset(pkg, oracle, fmsig):
    """Block sizes other than 65536 (within the supported range) and Reset() mid-stream."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    d = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    sizes = [65536, 8192, 10007, 65535, 32768, 12345, 65536, 9000, 65536, 65536]
    pos = 0
    for i, n in enumerate(sizes):
        if i == 6:
            o.reset()
            d.Reset()
        iq = fmsig.generate_f32(p, pos, n)
        pos += n
        r = o.process_stream(iq)
        a = d.ProcessStream(iq.view(np.complex64))
        assert a.shape == r.shape, (i, n)
        assert _bits_equal(a, r), (i, n, _rms(a, r))
    with pytest.raises(pkg.FmdError):
        d.ProcessStream(np.zeros(50, np.complex64))  # below fmd_batch_min_samples(): rejected loudly


def test_concurrency_modes_agree(pkg, fmsig):
    """Internal-stream execution (modes 1 and 2, incl. cross-call overlap) gives exactly the
    results of the fully serialized mode 0: audio bits and RDS groups, 4 channels, 40 calls."""
    import torch
    fs, D, C = 2.4e6, 11, 4
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=7 + c, pi=0x2000 + c, ps="MODE%04d" % c)
          for c in range(C)]
    nblk = 40
    iq = torch.from_numpy(np.stack([np.stack([fmsig.generate_f32(ps[c], b * N, N) for c in range(C)])
                                    for b in range(nblk)])).cuda()  # [nblk, C, 2N]
    stream = torch.cuda.current_stream().cuda_stream
    results = {}
    for mode in (0, 1, 2):
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C)
        b.set_concurrency(mode)
        stride = (b.max_audio_floats(N) + 3) // 4 * 4
        outs = [torch.zeros((C, stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
        groups, nfs = [], []
        for k in range(nblk):
            nfs.append(b.process_device(iq[k].data_ptr(), N, N, outs[k].data_ptr(), stride, stream))
            if mode == 2 and k >= 1:
                groups += b.collect_rds(stream=stream, lag=1)
            elif mode != 2:
                groups += b.collect_rds(stream=stream)
        b.wait(stream=stream)
        groups += b.collect_rds(stream=stream)
        torch.cuda.synchronize()
        results[mode] = (np.stack([outs[k].cpu().numpy()[:, :nfs[k]].copy() if nfs[k] == nfs[0]
                                   else np.pad(outs[k].cpu().numpy()[:, :nfs[k]], ((0, 0), (0, nfs[0] + 2 - nfs[k])))[:, :nfs[0]]
                                   for k in range(nblk)]), nfs, groups)
        b.close()
    a0, n0, g0 = results[0]
    assert len(g0) > 10
    for mode in (1, 2):
        a, n, g = results[mode]
        assert n == n0
        assert np.array_equal(a.view(np.uint32), a0.view(np.uint32)), mode
        assert g == g0, mode


DEGENERATE = ["silence", "dc", "noise", "clipped", "tone_at_carrier", "tiny"]


@pytest.mark.parametrize("kind", DEGENERATE)
def test_degenerate_inputs(pkg, oracle, kind):
    """Inputs no station produces: an unplugged antenna (all zeros: atan2f(0, 0), 0 / 0 behind the
    pilot loop's select), DC, white noise (PLLs never lock, the NCO limits and the rare-input paths
    are hit), full-scale square waves, an unmodulated carrier exactly at the tuned frequency, and
    a signal near the bottom of the float range.  Audio bits and the status getters equal the
    oracle's; all of it must stay finite."""
    fs, D = 2.4e6, 11
    rng = np.random.default_rng(12345)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    d = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    n0 = 0
    for blk in range(4):
        t = np.arange(n0, n0 + N, dtype=np.float64)
        if kind == "silence":
            z = np.zeros(N, np.complex64)
        elif kind == "dc":
            z = np.full(N, 0.3 + 0.2j, np.complex64)
        elif kind == "noise":
            z = (0.5 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
        elif kind == "clipped":
            z = (np.where((t // 7) % 2 == 0, 1.0, -1.0) + 1j * np.where((t // 5) % 2 == 0, 1.0, -1.0)).astype(np.complex64)
        elif kind == "tone_at_carrier":
            z = (0.5 * np.exp(2j * np.pi * (-0.15) * t)).astype(np.complex64)  # lands on 0 Hz after the tuner
        else:
            z = (1e-30 * np.exp(2j * np.pi * (-0.15 + 0.01 * np.sin(2 * np.pi * 1000 / fs * t)) * t)).astype(np.complex64)
        n0 += N
        iq = np.ascontiguousarray(z).view(np.float32)
        r = o.process_stream(iq)
        a = d.ProcessStream(z)
        assert np.all(np.isfinite(r)), (kind, blk)
        assert _bits_equal(a, r), (kind, blk, _rms(a, r))
    so = o.status()
    assert d.StereoDetected() == bool(so.stereo)
    assert np.float32(d.GetPilotLevel()) == np.float32(so.pilot_level)
    assert np.float32(d.GetBasebandLevel()) == np.float32(so.baseband_level)
    assert np.float32(d.GetInterfaceLevel()) == np.float32(so.if_level)
    assert np.float32(d.GetTuningOffset()) == np.float32(so.tuning_offset)


def test_degenerate_inputs_whole_cu_form(pkg, oracle):
    """The same six inputs through a 1026-channel batch with overlapped calls: the serial stage runs in its
    whole-CU form (hand-counted waits, per-pair hand-off, out-of-line rare-input path; one workgroup
    with an empty second group and padded lanes).  Channel c gets input c mod 6: one channel of
    each kind against the oracle, every other channel against its twin."""
    import torch
    fs, D, C, nblk = 2.4e6, 11, 1026, 3
    rng = np.random.default_rng(12345)
    base = np.zeros((nblk, 6, N), np.complex64)
    for blk in range(nblk):
        t = np.arange(blk * N, (blk + 1) * N, dtype=np.float64)
        base[blk, 1] = 0.3 + 0.2j
        base[blk, 2] = 0.5 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
        base[blk, 3] = np.where((t // 7) % 2 == 0, 1.0, -1.0) + 1j * np.where((t // 5) % 2 == 0, 1.0, -1.0)
        base[blk, 4] = 0.5 * np.exp(2j * np.pi * (-0.15) * t)
        base[blk, 5] = 1e-30 * np.exp(2j * np.pi * (-0.15 + 0.01 * np.sin(2 * np.pi * 1000 / fs * t)) * t)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    stride = (b.max_audio_floats(N) + 63) // 64 * 64
    st = torch.cuda.current_stream().cuda_stream
    idx = torch.arange(C, device="cuda") % 6
    iq = [torch.view_as_real(torch.from_numpy(base[k]).cuda()[idx]).contiguous() for k in range(nblk)]
    audio = [torch.zeros((C, stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    nf = [b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), stride, st) for k in range(nblk)]
    b.wait(stream=st)
    torch.cuda.synchronize()
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(6)]
    for k in range(nblk):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for kind in range(6):
            r = refs[kind].process_stream(np.ascontiguousarray(base[k, kind]).view(np.float32))
            assert _bits_equal(a[kind], r), (k, kind)
            twins = a[kind::6]
            assert np.array_equal(twins.view(np.uint32), np.broadcast_to(a[kind], twins.shape).view(np.uint32)), (k, kind)
    b.close()


def test_serial_stage_probe(pkg, fmsig):
    """The per-workgroup probe of the serial stage (dev aid): off by default, and with the
    "serial_probe" switch of fmd_batch_debug_set one (start, end, cycles) record per workgroup and
    launch, in both forms."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01)
    iq = fmsig.generate_f32(p, 0, N).view(np.complex64)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 2)
    b.process_host(np.stack([iq, iq]))
    assert b.debug_serial_probe().shape[1] == 0
    b.close()
    for C, wgs in ((2, 1), (1100, 9)):  # shared form: one workgroup per 64 channels; whole-CU form: per 128
        b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
        b.debug_set("serial_probe", 1)
        b.process_host(np.broadcast_to(iq, (C, N)).copy())
        b.process_host(np.broadcast_to(iq, (C, N)).copy())
        pr = b.debug_serial_probe()
        used = [pr[l][pr[l][:, 1] > 0] for l in range(8)]
        used = [u for u in used if len(u)]
        assert len(used) == 2 and all(len(u) == wgs for u in used), [len(u) for u in used]
        for u in used:
            ticks = u[:, 1] - u[:, 0]
            cycles = u[:, 2] & ((1 << 40) - 1)
            assert (ticks > 0).all() and (cycles > 100000).all()
            assert ((cycles / ticks) > 5).all() and ((cycles / ticks) < 40).all()  # shader clock / 100 MHz
        b.close()


def test_ragged_block_sizes_overlapped_calls(pkg, oracle, fmsig):
    """Block sizes other than 65536 with overlapped calls (the IF FIR then runs two tiles per
    workgroup, k_if_fir_mt; odd sizes leave a ragged last sample and a partial last tile)."""
    import torch
    fs, D, C = 2.4e6, 11, 3
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=21 + c) for c in range(C)]
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(C)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    st = torch.cuda.current_stream().cuda_stream
    sizes = [65536, 8193, 10007, 65535, 32768, 12345, 65536, 9001, 65536]
    stride = (b.max_audio_floats(N) + 3) // 4 * 4
    pos, ins, outs, nfs = 0, [], [], []
    for n in sizes:
        iq = np.stack([fmsig.generate_f32(ps[c], pos, n) for c in range(C)])
        pos += n
        ins.append(iq)
        d_iq = torch.zeros((C, 2 * N), dtype=torch.float32, device="cuda")  # rows N IQ samples apart
        d_iq[:, :2 * n] = torch.from_numpy(iq).cuda()
        d_out = torch.zeros((C, stride), dtype=torch.float32, device="cuda")
        nfs.append(b.process_device(d_iq.data_ptr(), N, n, d_out.data_ptr(), stride, st))
        outs.append((d_iq, d_out))
    b.wait(stream=st)
    torch.cuda.synchronize()
    for i, n in enumerate(sizes):
        a = outs[i][1].cpu().numpy()
        for c in range(C):
            r = refs[c].process_stream(ins[i][c])
            assert nfs[i] == r.size and _bits_equal(a[c, :nfs[i]], r), (i, n, c)
    b.close()


def test_generic_filter_kernels_bit_exact(pkg, oracle, fmsig):
    """The generic half-band and ring-FIR kernels (what geometries outside the unrolled kernels' range run:
    very short or very long half-bands, filters shorter than a group of outputs), forced here on the
    usual geometry ("hb4" / "ring4" = 0 of fmd_batch_debug_set): every stage tap bit-exact."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=19)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    b.debug_set("hb4", 0)
    b.debug_set("ring4", 0)
    b.debug_set("halfband_chain", 0)
    b.enable_taps()
    for blk, n in enumerate([N, N, 30000, N, 2000, N]):
        iq = fmsig.generate_f32(p, blk * N, n)
        a_ref = o.process_stream(iq)
        a_gpu = b.process_host(iq.view(np.complex64), shared=True)[0]
        taps = o.taps()
        for name in STAGES:
            assert _bits_equal(b.tap(name).view(np.float32), taps[name].view(np.float32)), (blk, name)
        assert _bits_equal(a_gpu, a_ref), blk
    b.close()
