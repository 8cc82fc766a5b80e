"""Random geometries through the BATCH dispatch against the oracle.

`tests/test_gpu_geometries.py` walks a hand-picked list; which kernel form a batch gets (IF FIR: one or two tiles per
workgroup, one / two outputs per lane, the long-filter loops by the parity of D; resampler: ring or window-per-wave;
serial stage: whole-CU or shared; stream layout by filter length; sub-batches; CIC stages) is decided from IF rate,
downsample, filter order, channel count, input format and concurrency mode together (`csrc/fmd_batch_if.inc.hpp`,
`fmd_batch_process.inc.hpp`), and a hand-picked list only visits the corners somebody thought of.  Points drawn with a
fixed seed (40 in the suite; 1000 once, docs/MEASUREMENTS.md): IF rate 0.25-16.2 MS/s, every downsample that leaves
180-420 kHz of baseband (and four geometries of 5.4-12 MHz: `CCicN3DecimateBy2`), IF filter orders from 2 to 4096
(`cDownsampleFilter` takes any: DownConvert.h:36-39), 1-8320 channels (not multiples of 64 among them; 8320 = two
sub-batches), float and byte input, tuner tables of 1 to 1000 entries (FmDecode.h:42), captures shared by
many channels, tuning anywhere in +-0.4 fs, PCM rate 44.1 / 48 / 96 kHz, bandwidth, 50 / 75 us, the three
concurrency modes, outputs consumed at once or a call late, ragged calls -- three channels bit for bit against the
oracle: audio, getters, groups.
"""
import os
import random

import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _case(i):
    r = random.Random(977 + i)
    cic = r.random() < 0.06  # baseband >= 5.33 MHz: CCicN3DecimateBy2 in front of the half-bands (DownConvert.cpp:690-727)
    if cic:
        fs, D = r.choice([(6.4e6, 1), (12e6, 1), (12e6, 2), (16.2e6, 3)])
        order = 0
        C = r.choice([1, 3, 64, 65, 130])
    else:
        fs = r.choice([250e3, 400e3, 1.0e6, 1.2e6, 1.44e6, 1.8e6, 2.048e6, 2.4e6, 2.56e6, 2.88e6, 3.2e6, 5.0e6, 8.0e6,
                       10e6])
        D = r.choice([d for d in range(1, 56) if 180e3 <= fs / d <= 420e3])
        order = r.choice([0, 0, 0, 4 * D, 16 * D, 100, 257, 512, 1000, 2048, 4096, 2, 3, 7])  # (1: refused, MakeLanczosCoeff divides by zero)
        C = r.choice([1, 3, 64, 65, 200, 1000, 1024, 1100, 2049, 4160, 4160, 8320])
    shared = (not cic) and r.random() < 0.25
    table = r.choice([64, 256, 256, 1, 2, 3, 1000]) if shared else r.choice([0, 0, 0, 1, 7, 100, 1000])
    u8 = r.random() < 0.4  # (a shared capture of bytes: one station's, the float captures are three summed)
    mode = r.choice([0, 1, 2, 2, 2])
    lag = r.choice([0, 1]) if mode == 2 else 0
    pcm = r.choice([48000.0, 48000.0, 48000.0, 44100.0, 96000.0])
    bw = r.choice([15000.0, 15000.0, 12000.0])
    us = r.random() < 0.25
    tune = 0.0 if shared else round(r.uniform(-0.4, 0.4), 4)
    nmax = min(N, 32700 * D)
    lo = max(2000, 8 * max(order, 8 * D))  # (well above fmd_batch_min_samples(); shorter calls: test_short_calls_bit_exact)
    even = 8 if cic else 1  # (the CIC stage reads past an odd block, DownConvert.cpp:701: such calls are refused)
    calls = [(nmax if r.random() < 0.5 else r.randrange(min(lo, nmax), nmax + 1)) // (even * D) * (even * D) if cic
             else (nmax if r.random() < 0.5 else r.randrange(min(lo, nmax), nmax + 1)) for _ in range(5)]
    return dict(fs=fs, D=D, order=order, C=C, shared=shared, table=table, u8=u8, mode=mode, lag=lag, pcm=pcm, bw=bw,
                us=us, tune=tune, calls=calls, seed=i)


CASES = [_case(i) for i in range(int(os.environ.get("FMD_FUZZ_CASES", "40")))]  # (docs/MEASUREMENTS.md: run once at 600)
IDS = ["%02d-%gM-D%d-o%d-C%d%s%s-m%d%s-pcm%g%s" % (k["seed"], k["fs"] / 1e6, k["D"], k["order"], k["C"],
                                                   ("-shared%d" if k["shared"] else "-t%d") % k["table"],
                                                   "-u8" if k["u8"] else "", k["mode"], "-lag1" if k["lag"] else "",
                                                   k["pcm"] / 1e3, "-us" if k["us"] else "") for k in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_random_geometry_bit_exact(oracle, fmsig, case):
    import torch
    pkg = load_package()
    fs, D, order, C = case["fs"], case["D"], case["order"], case["C"]
    shared, table, u8 = case["shared"], case["table"], case["u8"]
    r = random.Random(case["seed"])
    tune, pcm, bw, us, lag = case["tune"] * fs, case["pcm"], case["bw"], case["us"], case["lag"]
    shifts = np.array([r.randrange(-table, table + 1) for _ in range(C)], dtype=np.int32) if shared else None
    params = pkg.make_params(fs, tune, pcm, bw, D, us, table_size=table, if_filter_order=order)
    b = pkg.Batch(params, C, tuning_shifts=shifts, record_callbacks=False)
    b.set_concurrency(case["mode"])
    if shared:
        cpc = C if C <= 8192 else 64  # (a capture's channels may not straddle two sub-batches: 130 captures then)
        b.set_channels_per_capture(cpc)
        gen = fmsig.DeviceGenerator([fmsig.default_params(fs, f_offset=f0, amp=0.15, noise_sigma=0.004, seed=70 + j,
                                                          pi=0x6000 + j) for j, f0 in enumerate((-0.25 * fs, 0.0, 0.2 * fs))],
                                    "cuda")
        rows = C // cpc
    else:  # every channel its own station, at the tuned frequency
        plist = [fmsig.channel_params(fs, c) for c in range(C)]
        for p in plist:
            p.f_offset = tune
        gen = fmsig.DeviceGenerator(plist, "cuda")
        rows = C
    check = sorted({0, C // 2, C - 1})
    refs = {c: oracle.OracleDecoder(fs, tune, pcm, bw, D, us_version=us, table_size=table, if_filter_order=order,
                                    tuning_shift=int(shifts[c]) if shared else None) for c in check}
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(2)]
    iq = [torch.zeros((rows, N, 2), dtype=torch.uint8 if u8 else torch.float32, device="cuda") for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    pos, groups, nfs = 0, [], []

    def verify(k, lg):
        n = case["calls"][k]
        b.wait(stream=st, lag=lg)
        groups.append(b.collect_rds_array(cap=4 * C + 16, stream=st, lag=lg))
        torch.cuda.synchronize()
        a = audio[k % 2][:, :nfs[k]].cpu().numpy()
        for c in check:
            x = iq[k % 2][0 if shared else c, :n].cpu().numpy().reshape(-1)
            ref = refs[c].process_stream(fmsig.u8_to_f32(x) if u8 else x)
            assert ref.size == nfs[k] and np.array_equal(a[c].view(np.uint32), ref.view(np.uint32)), (k, c, n)
            if lg == 0:  # (the getters follow the newest call)
                so, sg = refs[c].status(), b.status(c)
                assert sg.stereo_detected == so.stereo, (k, c)
                for f_o, f_g in ((so.if_level, sg.interface_level), (so.baseband_level, sg.baseband_level),
                                 (so.pilot_level, sg.pilot_level), (so.tuning_offset, sg.tuning_offset)):
                    assert np.float32(f_o).view(np.uint32) == np.float32(f_g).view(np.uint32), (k, c)

    for k, n in enumerate(case["calls"]):
        if shared and u8:
            g3 = torch.zeros((3, n, 2), dtype=torch.uint8, device="cuda")
            gen.generate(g3, pos, n)
            iq[k % 2][:, :n] = g3[1]
        elif shared:
            g3 = torch.zeros((3, n, 2), dtype=torch.float32, device="cuda")
            gen.generate(g3, pos, n)
            iq[k % 2][:, :n] = g3.sum(dim=0)  # (three stations summed into the one capture; every capture the same)
        else:
            blk = torch.zeros((C, n, 2), dtype=iq[0].dtype, device="cuda")
            gen.generate(blk, pos, n)
            iq[k % 2][:, :n] = blk
            del blk
        torch.cuda.synchronize()
        nfs.append(b.process_device(iq[k % 2].data_ptr(), N, n, audio[k % 2].data_ptr(), a_stride, st, u8=u8))
        if k >= lag:
            verify(k - lag, lag)
        pos += n
    for k in range(len(case["calls"]) - lag, len(case["calls"])):
        verify(k, 0)
    g = np.concatenate(groups)
    for c in check:
        mine = sorted((int(k), tuple(int(v) for v in bl)) for ch, k, bl in zip(g["channel"], g["call_index"], g["blocks"])
                      if ch == c)
        assert mine == sorted((k, tuple(bl)) for k, bl in refs[c].rds_groups()), c
    b.close()


def _drive(pkg, torch, oracle, fmsig, fs, D, C, calls, seed, stream, results, key, barrier=None):
    """One overlapped batch, outputs consumed a call late, three channels against the oracle; errors into results."""
    try:
        with torch.cuda.stream(stream):
            st = stream.cuda_stream
            gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c, base_seed=seed) for c in range(C)], "cuda")
            b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
            b.set_concurrency(2)
            check = [0, C // 2, C - 1]
            refs = {c: oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for c in check}
            a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
            audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(2)]
            iq = [torch.zeros((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(2)]
            nfs, pos = [], 0

            def verify(k, lg):
                b.wait(stream=st, lag=lg)
                b.collect_rds_array(cap=4 * C, stream=st, lag=lg)
                stream.synchronize()
                a = audio[k % 2][:, :nfs[k]].cpu().numpy()
                for c in check:
                    ref = refs[c].process_stream(iq[k % 2][c, :calls[k]].cpu().numpy().reshape(-1))
                    assert ref.size == nfs[k] and np.array_equal(a[c].view(np.uint32), ref.view(np.uint32)), (key, k, c)

            for k, n in enumerate(calls):
                blk = torch.zeros((C, n, 2), dtype=torch.float32, device="cuda")
                gen.generate(blk, pos, n)
                iq[k % 2][:, :n] = blk
                stream.synchronize()
                if barrier is not None:
                    barrier.wait()  # both threads submit at the same moment
                nfs.append(b.process_device(iq[k % 2].data_ptr(), N, n, audio[k % 2].data_ptr(), a_stride, st))
                if k >= 1:
                    verify(k - 1, 1)
                pos += n
            verify(len(calls) - 1, 0)
            b.close()
        results[key] = "ok"
    except BaseException as e:  # noqa: BLE001 -- reported by the test's thread
        results[key] = repr(e)
        if barrier is not None:
            barrier.abort()


def test_two_batches_of_different_geometry_from_two_threads(oracle, fmsig):
    """Two decoders of different geometry in one process, each driven by a thread of its own on a stream of its own,
    their overlapped calls submitted at the same moment (a barrier in front of every call): the library keeps no
    state between decoders (the reference's one process-wide static, `ps_text`, RDSGroupDecoder.cpp:311, is
    per-decoder here) -- each batch's audio is the oracle's, bit for bit."""
    import threading

    import torch
    pkg = load_package()
    torch.cuda.synchronize()
    results = {}
    barrier = threading.Barrier(2, timeout=120)
    calls = [N, 30001, N, 12346, N, N, 2500, N]
    specs = [("2.4M", 2.4e6, 11, 1024, 5000), ("1.0M", 1.0e6, 4, 1100, 9000)]
    threads = [threading.Thread(target=_drive, args=(pkg, torch, oracle, fmsig, fs, D, C, calls, seed,
                                                     torch.cuda.Stream(priority=-1), results, key, barrier))
               for key, fs, D, C, seed in specs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert results == {"2.4M": "ok", "1.0M": "ok"}, results
