"""The other BASELINE configs as parity cases, and the C++ drop-in class used from a program
that links only libfmd_hip.so (no torch in the process)."""
import os
import subprocess

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_config3_many_channels_from_one_capture(oracle, fmsig):
    """BASELINE config 3: channels freq-shifted from ONE 2.4 MS/s capture (shared input, read
    once per tile by every channel).  cFineTuner table_size = 256 (ctor parameter,
    FmDecode.h:42) so shifts are k * 9.375 kHz; 256 channels, 8 of them checked bit for bit
    against the oracle with the same shift."""
    pkg = load_package()
    fs, D, C, T = 2.4e6, 11, 256, 256
    stations = [fmsig.default_params(fs, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i,
                                     pi=0x5000 + i, ps="CAP%05d" % i, f_left=500.0 + 300 * i)
                for i, f0 in enumerate((-600e3, -360e3, -150e3, 75e3, 300e3, 600e3))]
    shifts = np.arange(C, dtype=np.int32) - 128
    b = pkg.Batch(pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T), C, tuning_shifts=shifts)
    check = [0, 64, 90, 112, 136, 160, 192, 255]  # 64 -> -600 kHz ... 192 -> +600 kHz
    refs = {c: oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T,
                                    tuning_shift=int(shifts[c])) for c in check}
    for blk in range(10):
        cap = np.zeros(2 * N, dtype=np.float32)
        for p in stations:
            cap += fmsig.generate_f32(p, blk * N, N)
        audio = b.process_host(cap.view(np.complex64), shared=True)
        for c in check:
            r = refs[c].process_stream(cap)
            assert _bits_equal(audio[c], r), (blk, c)
    # channel 64 (shift -64 * 9375 Hz = -600 kHz) brings the +600 kHz station to 0: it sees a pilot,
    # channel 0 (tuned 1.2 MHz away from everything) does not
    assert b.status(64).pilot_level > 0.08 > abs(b.status(0).pilot_level)


@pytest.mark.parametrize("device_call", [False, True], ids=["host-buffers", "device-buffers-overlapped"])
def test_config3_scaled_out_several_captures_in_one_batch(oracle, fmsig, device_call):
    """Config 3 scaled out ("a shared capture"): G captures x k stations each in ONE batch
    (fmd_batch_set_channels_per_capture) -- channels [g k, (g + 1) k) tune capture g, the process calls take one
    input row per capture.  4 captures x 64 channels = 256 here, through the host-buffer call and through
    device-resident overlapped calls with ragged sizes; per capture several channels bit for bit against the
    oracle fed THAT capture with the same shift (the only stage that sees the capture is the tuner,
    FmDecode.cpp:66-82)."""
    pkg = load_package()
    fs, D, G, k, T = 2.4e6, 11, 4, 64, 256
    C = G * k
    offs = (-600e3, -360e3, -150e3, 75e3, 300e3, 600e3)
    caps = [[fmsig.default_params(fs, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i + 16 * g,
                                  pi=0x5000 + i + 16 * g, ps="CAP%05d" % (i + 16 * g), f_left=500.0 + 300 * i + 7 * g)
             for i, f0 in enumerate(offs)] for g in range(G)]
    shifts = (np.arange(C, dtype=np.int32) % k) * 4 - 128  # every capture: 64 shifts across the band
    b = pkg.Batch(pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T), C, tuning_shifts=shifts,
                  record_callbacks=False)
    b.set_channels_per_capture(k)
    check = [0, 16, 63, 64, 80, 127, 128 + 48, 255]
    refs = {c: oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T, tuning_shift=int(shifts[c]))
            for c in check}
    sizes = [N, 30001, N, 12346, N, N] if device_call else [N] * 5
    pos, blocks = 0, []
    for n in sizes:
        cap = np.zeros((G, 2 * n), dtype=np.float32)
        for g in range(G):
            for p in caps[g]:
                cap[g] += fmsig.generate_f32(p, pos, n)
        blocks.append(cap)
        pos += n
    if not device_call:
        for blk, cap in enumerate(blocks):
            audio = b.process_host(cap.view(np.complex64))
            for c in check:
                assert _bits_equal(audio[c], refs[c].process_stream(cap[c // k])), (blk, c)
    else:
        import torch
        b.set_concurrency(2)
        a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
        st = torch.cuda.current_stream().cuda_stream
        d_iq, d_audio, nf = [], [], []
        for cap in blocks:
            n = cap.shape[1] // 2
            n_al = (n + 1) // 2 * 2
            t = torch.zeros((G, n_al, 2), dtype=torch.float32, device="cuda")
            t[:, :n] = torch.from_numpy(cap.reshape(G, n, 2)).cuda()
            d_iq.append((t, n, n_al))
            d_audio.append(torch.zeros((C, a_stride), dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
        for i, (t, n, n_al) in enumerate(d_iq):
            nf.append(b.process_device(t.data_ptr(), n_al, n, d_audio[i].data_ptr(), a_stride, st))
        b.wait(stream=st)
        torch.cuda.synchronize()
        for blk, cap in enumerate(blocks):
            a = d_audio[blk][:, :nf[blk]].cpu().numpy()
            for c in check:
                assert _bits_equal(a[c], refs[c].process_stream(cap[c // k])), (blk, c)
    # capture g's station at +600 kHz is brought to 0 by shift -64: channel g k + 16 sees a pilot
    assert all(b.status(g * k + 16).pilot_level > 0.08 for g in range(G))
    b.close()


def test_several_captures_of_rtl_sdr_bytes(oracle, fmsig):
    """The same with byte input (fmd_batch_process_host_u8: one row of RTL-SDR byte pairs per capture, converted
    inside the IF kernel like ReadAsyncCB, RTL_SDR_Source.cpp:207-211): 2 captures x 64 channels."""
    pkg = load_package()
    fs, D, G, k, T = 2.4e6, 11, 2, 64, 256
    C = G * k
    shifts = (np.arange(C, dtype=np.int32) % k) * 4 - 128
    b = pkg.Batch(pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T), C, tuning_shifts=shifts,
                  record_callbacks=False)
    b.set_channels_per_capture(k)
    check = [0, 16, 63, 64, 80, 127]
    refs = {c: oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T, tuning_shift=int(shifts[c]))
            for c in check}
    ps = [fmsig.default_params(fs, f_offset=-600e3, noise_sigma=0.01, seed=70 + g, pi=0x6000 + g) for g in range(G)]
    pos = 0
    for blk, n in enumerate([N, 30002, N, 1000, N]):
        cap = np.stack([fmsig.generate_u8(p, pos, n) for p in ps])  # [G][2 n] bytes
        audio = b.process_host_u8(cap)
        for c in check:
            assert _bits_equal(audio[c], refs[c].process_stream_u8(cap[c // k])), (blk, c)
        pos += n
    b.close()


def test_config5_long_fir_10msps(oracle, fmsig):
    """BASELINE config 5 geometry: 4096-tap cDownsampleFilter at 10 MS/s, D = 46."""
    pkg = load_package()
    fs, D, C = 10e6, 46, 3
    ps = [fmsig.default_params(fs, noise_sigma=0.01, seed=900 + c, pi=0x7000 + c) for c in range(C)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=4096), C)
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=4096)
            for _ in range(C)]
    b.enable_taps()
    for blk in range(4):
        iq = np.stack([fmsig.generate_f32(ps[c], blk * N, N) for c in range(C)])
        audio = b.process_host(iq.view(np.complex64).reshape(C, N))
        for c in range(C):
            r = refs[c].process_stream(iq[c])
            assert _bits_equal(b.tap("demod", c), refs[c].taps()["demod"]), (blk, c)
            assert _bits_equal(audio[c], r), (blk, c)


def test_cpp_class_drop_in_without_torch(tmp_path, oracle, fmsig):
    """tests/cpp/receiver_demo.cpp uses cFmDecoder exactly like cRadioReceiver does and links
    only libfmd_hip.so; its audio and byte-stuffed UECP stream equal the oracle's."""
    pkg = load_package()
    exe = os.path.join(ROOT, "tests", "cpp", "receiver_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pvr.rtl.radiofm_amd", "csrc")])
    fs, D, nblk = 2.4e6, 11, 40
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=3)
    iq = np.concatenate([fmsig.generate_f32(p, b * N, N) for b in range(nblk)])
    iq_path, a_path, u_path = tmp_path / "iq.f32", tmp_path / "audio.f32", tmp_path / "uecp.bin"
    iq.tofile(iq_path)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    out = subprocess.run([exe, str(iq_path), str(fs), str(D), str(a_path), str(u_path)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    ref = np.concatenate([o.process_stream(iq[2 * N * b:2 * N * (b + 1)]) for b in range(nblk)])
    got = np.fromfile(a_path, dtype=np.float32)
    assert _bits_equal(got, ref)
    stuffed = b"".join(pkg.stuff_uecp_frame(f) for f in o.uecp_frames())
    assert u_path.read_bytes() == stuffed and len(stuffed) > 0
    assert "stereo=1" in out.stdout and "name=TESTFM01" in out.stdout


def test_config4_full_shard_properties(oracle, fmsig):
    """BASELINE configs[3] at its full per-GPU size: 8192 channels x 65536 IQ per call, device-
    generated input, calls overlapped exactly like bench.py (concurrency 2, outputs consumed two
    calls late).  Too big to run every channel through the oracle, so: (1) 8 channels incl. the
    first and the last are checked against the oracle bit for bit on the very bytes the device
    generator produced; (2) size-independent property over the whole batch: channels c and
    c + 4096 are given the same station, so their audio and RDS groups must be identical."""
    import torch
    pkg = load_package()
    fs, D, C, nblk, LAG = 2.4e6, 11, 8192, 8, 2
    chans = [fmsig.channel_params(fs, c % 4096) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    for k in range(nblk):
        gen.generate(iq[k], k * N, N)
    st = torch.cuda.current_stream().cuda_stream
    nf, groups = [], []
    for k in range(nblk):
        nf.append(b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), a_stride, st))
        if k >= LAG:
            b.wait(stream=st, lag=LAG)
            groups.append(b.collect_rds_array(cap=4 * C, stream=st, lag=LAG))
    b.wait(stream=st)
    groups.append(b.collect_rds_array(cap=4 * C, stream=st))
    torch.cuda.synchronize()
    check = [0, 1, 63, 64, 4095, 4096, 8190, 8191]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
    for k in range(nblk):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for c in check:
            r = refs[c].process_stream(iq[k][c].cpu().numpy().reshape(-1))
            assert _bits_equal(a[c], r), (k, c)
        assert np.array_equal(a[:4096].view(np.uint32), a[4096:].view(np.uint32)), k
    g = np.concatenate(groups)
    lo, hi = g[g["channel"] < 4096], g[g["channel"] >= 4096]
    key = lambda x, off: sorted((int(c) - off, int(k), tuple(int(v) for v in bl))
                                for c, k, bl in zip(x["channel"], x["call_index"], x["blocks"]))
    assert key(lo, 0) == key(hi, 4096)
    for c in check:
        mine = sorted((int(k), tuple(int(v) for v in bl)) for ch, k, bl in
                      zip(g["channel"], g["call_index"], g["blocks"]) if ch == c)
        assert mine == sorted((k, tuple(bl)) for k, bl in refs[c].rds_groups()), c
    b.close()


def test_config5_full_size_against_oracle(oracle, fmsig):
    """BASELINE configs[4] at its full size: 4096 channels @10 MS/s, D = 46, 4096-tap IF FIR, 65536 IQ
    per call, device-generated input, calls overlapped like bench.py.  8 channels incl. the first and
    the last against the oracle bit for bit on the very bytes the device generator produced (the
    oracle's cDownsampleFilter takes any order, like the reference's: DownConvert.h:36-39), and the
    size-independent property over the whole batch: channels c and c + 2048 carry the same station."""
    import torch
    pkg = load_package()
    fs, D, order, C, nblk, LAG = 10e6, 46, 4096, 4096, 5, 2
    chans = [fmsig.channel_params(fs, c % 2048) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, "cuda")
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order), C,
                  record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    for k in range(nblk):
        gen.generate(iq[k], k * N, N)
    st = torch.cuda.current_stream().cuda_stream
    nf = []
    for k in range(nblk):
        nf.append(b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), a_stride, st))
        if k >= LAG:
            b.wait(stream=st, lag=LAG)
    b.wait(stream=st)
    torch.cuda.synchronize()
    check = [0, 1, 63, 64, 2047, 2048, 4094, 4095]
    refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order) for c in check}
    for k in range(nblk):
        a = audio[k][:, :nf[k]].cpu().numpy()
        for c in check:
            r = refs[c].process_stream(iq[k][c].cpu().numpy().reshape(-1))
            assert _bits_equal(a[c], r), (k, c)
        assert np.array_equal(a[:2048].view(np.uint32), a[2048:].view(np.uint32)), k
    for c in check:
        so, sg = refs[c].status(), b.status(c)
        assert sg.stereo_detected == so.stereo
        assert np.float32(sg.pilot_level) == np.float32(so.pilot_level)
    b.close()


def test_seek_stops_on_the_next_stereo_station(oracle, fmsig):
    """SURVEY 8(f)-4, the consumer of the getters: the tuner dialog's seek (cChannelSettings::Process,
    /root/reference/src/ChannelSettings.cpp:96-147) steps the tuned frequency by 100 kHz, listens for 1.25 s
    and stops when GetSignalStatus reports stereo (RadioReceiver.cpp:544-572).  One capture holds the whole
    2.4 MHz around the tuner: a batch with one channel per 100 kHz step (cFineTuner table of 24 entries,
    FmDecode.cpp:45-58) listens to all 24 steps at once -- the seek's answer from any start, in one dwell.  Checked
    against the reference's way: an oracle decoder per step, each fed the same 1.25 s."""
    pkg = load_package()
    fs, D, T = 2.4e6, 11, 24
    # three stereo stations and a mono one on the 100 kHz grid, the other steps empty
    stations = [fmsig.default_params(fs, f_offset=-700e3, amp=0.2, noise_sigma=0.004, seed=81, pi=0x7001),
                fmsig.default_params(fs, f_offset=-200e3, amp=0.2, noise_sigma=0.004, seed=82, pi=0x7002),
                fmsig.default_params(fs, f_offset=500e3, amp=0.2, noise_sigma=0.004, seed=83, pi=0x7003),
                fmsig.mono_params(fs, f_offset=100e3, amp=0.2, noise_sigma=0.004, seed=84)]
    shifts = np.arange(T, dtype=np.int32) - 12           # tuned to -1.2 MHz ... +1.1 MHz in 100 kHz steps
    b = pkg.Batch(pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T), T, tuning_shifts=shifts,
                  record_callbacks=False)
    refs = [oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T, tuning_shift=int(k)) for k in shifts]
    blocks = int(np.ceil(1.25 * fs / N))                  # the dialog's dwell
    for blk in range(blocks):
        cap = np.zeros(2 * N, dtype=np.float32)
        for p in stations:
            cap += fmsig.generate_f32(p, blk * N, N)
        b.process_host(cap.view(np.complex64), shared=True)
        for r in refs:
            r.process_stream(cap)
    stereo = [bool(b.status(c).stereo_detected) for c in range(T)]
    stereo_ref = [bool(r.status().stereo) for r in refs]
    assert stereo == stereo_ref
    # a station at offset f is brought to 0 by shift -f / 100 kHz: channel index 12 - f / 100 kHz.  The pilot PLL
    # also locks one step beside a stereo station (the decoder's behaviour, the reference's seek stops there too);
    # the mono station (11) and its neighbours are passed, the empty steps too
    on = [c for c in range(T) if stereo[c]]
    assert {7, 14, 19} <= set(on) <= {6, 7, 8, 13, 14, 15, 18, 19, 20} and not ({10, 11, 12} & set(on))

    def seek(flags, start, step):  # cChannelSettings::Process: step until stereo (wrapping like its band limits)
        c = start
        for _ in range(T):
            c = (c + step) % T
            if flags[c]:
                return c
        return None
    for start in range(T):
        for step in (+1, -1):
            assert seek(stereo, start, step) == seek(stereo_ref, start, step)
    assert seek(stereo, 9, +1) in (13, 14) and seek(stereo, 12, -1) in (8, 7)
    for c in (7, 14, 19):  # the getters the dialog shows (IF level, pilot) agree bit for bit with the reference path
        so, sg = refs[c].status(), b.status(c)
        assert np.float32(so.if_level) == np.float32(sg.interface_level)
        assert np.float32(so.pilot_level) == np.float32(sg.pilot_level)
    b.close()
