"""Long runs at the BASELINE configurations' full per-GPU sizes, every channel of every call compared between SCHEDULES.

The short parity tests compare a handful of channels with the oracle over 8-11 calls.  What they cannot see is a
cross-stream ordering mistake that only bites once in a few hundred calls, on a few channels, when the pipeline's
timing shifts (five internal streams, events in rotating slots, buffers doubled by call parity, host-side decisions
that depend on what has already completed: the input-ready event a call may skip, the waits `fmd_batch_wait_lagged`
may skip).  The decoder is deterministic: the same inputs give the same bits whatever the schedule.  So: a few hundred
calls under each of several schedules -- overlapped in the three stream layouts, consumed 1, 2 or 3 calls late, with
a host that stalls at random, on a stream of the caller's own, ordered after every call, strictly serial (concurrency
0: one stream, the reference's stage order), and as the two sub-batches of a 16 384-channel batch -- and every call's
audio (a position-weighted 64-bit checksum over all channels and frames, computed on the device), every RDS group and
the final status records must be THE SAME in all of them.  Against the oracle itself: four channels of the base run,
first calls (`ProcessStream`, FmDecode.cpp:417-501); the rest of the run hangs on them by determinism.
"""
import os
import random
import time

import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536
RING = 10  # (as bench.py: a shorter ring's seams come before the RDS decoder has found block sync)
SCALE = int(os.environ.get("FMD_SOAK_SCALE", "1"))  # calls per schedule x SCALE (docs/MEASUREMENTS.md: run at 10)

OVERLAPPED = {
    "light streams, 3 late": dict(concurrency=2, lag=3),
    "light streams, 2 late": dict(concurrency=2, lag=2),
    "light streams, 1 late": dict(concurrency=2, lag=1),
    "heavy stream": dict(concurrency=2, lag=3, debug=(("lpf_late", 0),)),
    "own stream": dict(concurrency=2, lag=2, debug=(("lpf_late", 1),)),
    "light streams forced": dict(concurrency=2, lag=2, debug=(("lpf_late", 2),)),
    "host stalls": dict(concurrency=2, lag=3, stalls=True),
    "host stalls, heavy stream, caller's stream": dict(concurrency=2, lag=2, stalls=True, own_stream=True,
                                                        debug=(("lpf_late", 0),)),
    "caller's stream": dict(concurrency=2, lag=3, own_stream=True),
    "ordered after every call": dict(concurrency=1, lag=0),
    "groups by device export": dict(concurrency=2, lag=3, export=True),
}


def _weights(torch, C, stride):
    g = torch.Generator(device="cuda")
    g.manual_seed(12345)
    return torch.randint(1, 1 << 20, (C, stride), dtype=torch.int64, device="cuda", generator=g)


def _a_stride(pkg, params):
    probe = pkg.Batch(params, 64, record_callbacks=False)
    s = (probe.max_audio_floats(N) + 63) // 64 * 64
    probe.close()
    return s


def _run(torch, make_batch, iq, iq_stride, w, *, concurrency, lag, calls, debug=(), stalls=False, own_stream=False,
         channel0=0, keep_audio=0, u8=False, sizes=None, reset_at=(), export=False):
    """One schedule over `calls` calls of the ring `iq`.  Returns (per-call checksums of channels
    [channel0, channel0 + w.shape[0]), their groups sorted, status tuples of four of them, the first `keep_audio`
    calls' audio of those four, frames per call).  sizes: IQ samples of call i (default: full blocks); reset_at: calls
    in front of which the batch is reset (`cFmDecoder::Reset`, FmDecode.cpp:326-338: a drain, then every channel);
    export: the groups leave through fmd_batch_export_rds_device (records in device memory, what the gather sends)
    instead of the host drain."""
    b = make_batch()
    C = b.n_channels
    b.set_concurrency(concurrency)
    for k, v in debug:
        b.debug_set(k, v)
    a_stride = w.shape[1]
    assert a_stride >= b.max_audio_floats(N)
    nbuf = lag + 2
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    cur = torch.cuda.Stream(priority=-1) if own_stream else torch.cuda.current_stream()
    st = cur.cuda_stream
    rnd = random.Random(99)
    torch.cuda.synchronize()
    CW = w.shape[0]
    sums, nf, groups, kept = [], [], [], []
    rec = torch.zeros((4 * C, 4), dtype=torch.int32, device="cuda") if export else None
    pkg_dtype = np.dtype([("channel", "<u4"), ("call_index", "<u4"), ("blocks", "<u2", (4,))])
    done_upto = 0
    picks = [0, 63, CW // 2, CW - 1]

    def finalize(i, lg):
        assert len(sums) == i
        b.wait(stream=st, lag=lg)
        if export:
            assert not b.export_rds_device(rec.data_ptr(), 4 * C, channel_offset=7, stream=st, lag=lg)
            r = rec.cpu().numpy()  # (stream-ordered behind the export kernel)
            r = r[r[:, 0] != 0]
            g = np.zeros(len(r), dtype=pkg_dtype)
            g["channel"] = r[:, 0] - 8
            g["call_index"] = r[:, 1]
            g["blocks"] = np.stack([r[:, 2] & 0xffff, (r[:, 2] >> 16) & 0xffff, r[:, 3] & 0xffff,
                                    (r[:, 3] >> 16) & 0xffff], axis=1)
        else:
            g = b.collect_rds_array(cap=4 * C, stream=st, lag=lg)
        groups.append(g[(g["channel"] >= channel0) & (g["channel"] < channel0 + CW)])
        a = audio[i % nbuf][channel0:channel0 + CW].view(torch.int32).to(torch.int64)
        sums.append((a[:, :nf[i]] * w[:, :nf[i]]).sum())
        if i < keep_audio:
            kept.append(audio[i % nbuf][channel0:channel0 + CW][picks, :nf[i]].clone())
        audio[i % nbuf].zero_()  # a call that did not write its audio shows

    with torch.cuda.stream(cur):
        for i in range(calls):
            if i in reset_at:
                for j in range(len(sums), i):  # (what the host has not consumed yet)
                    finalize(j, i - 1 - j)
                done_upto = i
                b.reset()
            n = sizes[i % len(sizes)] if sizes else N
            nf.append(b.process_device(iq[i % RING].data_ptr(), iq_stride, n, audio[i % nbuf].data_ptr(), a_stride,
                                       st, u8=u8))
            if i - lag >= done_upto:
                finalize(i - lag, lag)
            if stalls:
                r = rnd.random()
                if r < 0.08:
                    time.sleep(rnd.random() * 0.006)  # the pipeline runs dry
                elif r < 0.12:
                    torch.cuda.synchronize()
                elif r < 0.16:
                    cur.synchronize()
        for i in range(len(sums), calls):  # (lag 0: nothing left)
            finalize(i, calls - 1 - i)
    torch.cuda.synchronize()
    assert not b.take_rds_lost()
    status = []
    for c in picks:
        s = b.status(channel0 + c)
        status.append(tuple(getattr(s, f) for f in ("stereo_detected", "tuning_offset", "interface_level",
                                                      "baseband_level", "pilot_level", "rds_state")))
    b.close()
    g = np.concatenate(groups)
    order = np.lexsort((g["blocks"][:, 3], g["blocks"][:, 2], g["blocks"][:, 1], g["blocks"][:, 0],
                        g["call_index"], g["channel"]))
    g = g[order]
    recs = np.column_stack([g["channel"].astype(np.int64) - channel0, g["call_index"].astype(np.int64),
                            g["blocks"].astype(np.int64)])
    return torch.stack(sums).cpu().numpy(), recs, status, [k.cpu().numpy() for k in kept], nf


def _same(name, got, base):
    sums, recs, status, _, nf = got
    sums0, recs0, status0, _, nf0 = base
    assert nf == nf0, name
    bad = np.nonzero(sums != sums0)[0]
    assert bad.size == 0, "%s: audio of call(s) %s differs from the base run" % (name, bad[:8].tolist())
    assert np.array_equal(recs, recs0), "%s: RDS groups differ from the base run" % name
    assert status == status0, name


def _against_oracle(base, refs, rows):
    """refs: one oracle decoder per kept channel; rows(i): their input rows of call i (numpy, interleaved f32)."""
    _, _, _, kept, nf0 = base
    for i in range(len(kept)):
        x = rows(i)
        for j, ref in enumerate(refs):
            r = ref.process_stream(x[j])
            assert r.size == nf0[i] and np.array_equal(kept[i][j].view(np.uint32), r.view(np.uint32)), (j, i)


def test_config4_every_schedule_the_same_bits_over_300_calls(oracle, fmsig):
    """BASELINE configs[3]'s per-GPU shard, 8192 channels x 65536 IQ, 300 calls a schedule; then the same 8192
    channels twice in ONE 16 384-channel batch (two sub-batches on the shared streams), each half the same bits."""
    import torch
    pkg = load_package()
    fs, D, C, calls = 2.4e6, 11, 8192, 300 * SCALE
    params = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D)
    gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c % 4096) for c in range(C)], "cuda")
    # (the ring of the 16 384-channel batch; the 8192-channel runs take its first half: 80 GiB of the 288)
    big = [torch.empty((2 * C, N, 2), dtype=torch.float32, device="cuda") for _ in range(RING)]
    iq = [blk[:C] for blk in big]
    for k in range(RING):
        gen.generate(iq[k], k * N, N)
        big[k][C:].copy_(iq[k])
    w = _weights(torch, C, _a_stride(pkg, params))
    make = lambda: pkg.Batch(params, C, record_callbacks=False)

    base = _run(torch, make, iq, N, w, concurrency=0, lag=0, calls=calls, keep_audio=6)
    recs0 = base[1]
    assert len(recs0) > 20000
    picks = [0, 63, C // 2, C - 1]
    _against_oracle(base, [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in picks],
                    lambda i: [iq[i % RING][c].cpu().numpy().reshape(-1) for c in picks])
    # channels c and c + 4096 carry the same station: the same groups
    lo, hi = recs0[recs0[:, 0] < 4096], recs0[recs0[:, 0] >= 4096].copy()
    hi[:, 0] -= 4096
    assert np.array_equal(lo, hi)

    schedules = dict(OVERLAPPED)
    schedules["two post streams"] = dict(concurrency=2, lag=3, debug=(("split_post", 1),))
    for name, kw in schedules.items():
        _same(name, _run(torch, make, iq, N, w, calls=calls, **kw), base)
    # ragged calls (fewer audio frames than filter taps, odd lengths, full blocks) and two resets under way
    sizes = [N, 30001, N, 1001, N, N, 330, 65535, N, 12346, N, N, N, 150, N, 33001]
    rkw = dict(calls=calls // 2, sizes=sizes, reset_at=(calls // 6, calls // 3 + 1))
    rbase = _run(torch, make, iq, N, w, concurrency=0, lag=0, **rkw)
    for name in ("light streams, 3 late", "light streams, 1 late", "heavy stream", "own stream", "host stalls",
                 "host stalls, heavy stream, caller's stream"):
        _same("ragged, " + name, _run(torch, make, iq, N, w, **rkw, **OVERLAPPED[name]), rbase)
    make_big = lambda: pkg.Batch(params, 2 * C, record_callbacks=False)
    for ch0 in (0, C):
        _same("sub-batch at channel %d" % ch0,
              _run(torch, make_big, big, N, w, concurrency=2, lag=3, calls=calls, stalls=ch0 == C, channel0=ch0), base)
    _same("ragged, sub-batch at channel %d" % C,
          _run(torch, make_big, big, N, w, concurrency=2, lag=2, channel0=C, **rkw), rbase)


@pytest.mark.parametrize("mode", ["u8", "fma-waived", "shuffle-waived"])
def test_config4_other_modes_every_schedule_the_same_bits(oracle, fmsig, mode):
    """The same at 8192 channels for RTL-SDR byte input (`ReadAsyncCB`'s conversion inside the IF kernel,
    RTL_SDR_Source.cpp:196-213; bit-exact: against the oracle on the converted bytes) and for the two parity-waived
    reductions of the IF FIR (FMD_FIR_FMA_PARITY_WAIVED, FMD_FIR_SHUFFLE_PARITY_WAIVED: not the reference's bits,
    but still the same bits on every schedule -- base = the run ordered after every call)."""
    import torch
    pkg = load_package()
    fs, D, C, calls = 2.4e6, 11, 8192, 200 * SCALE
    u8 = mode == "u8"
    red = {"u8": 0, "fma-waived": pkg.FIR_FMA_PARITY_WAIVED, "shuffle-waived": pkg.FIR_SHUFFLE_PARITY_WAIVED}[mode]
    params = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, fir_reduction=red)
    gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c % 4096) for c in range(C)], "cuda")
    iq = [torch.empty((C, N, 2), dtype=torch.uint8 if u8 else torch.float32, device="cuda") for _ in range(RING)]
    for k in range(RING):
        gen.generate(iq[k], k * N, N)
    w = _weights(torch, C, _a_stride(pkg, params))
    make = lambda: pkg.Batch(params, C, record_callbacks=False)
    if u8:
        base = _run(torch, make, iq, N, w, concurrency=0, lag=0, calls=calls, keep_audio=6, u8=True)
        picks = [0, 63, C // 2, C - 1]
        _against_oracle(base, [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in picks],
                        lambda i: [fmsig.u8_to_f32(iq[i % RING][c].cpu().numpy().reshape(-1)) for c in picks])
    else:
        base = _run(torch, make, iq, N, w, concurrency=1, lag=0, calls=calls)
        # the check has teeth: the parity mode's audio (~1e-5 RMS away) is another checksum in every call
        exact = _run(torch, lambda: pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C,
                                              record_callbacks=False), iq, N, w, concurrency=1, lag=0, calls=12)
        assert (exact[0] != base[0][:12]).all()
    assert len(base[1]) > 10000
    for name, kw in OVERLAPPED.items():
        _same(name, _run(torch, make, iq, N, w, calls=calls, u8=u8, **kw), base)


def test_config5_every_schedule_the_same_bits(oracle, fmsig):
    """BASELINE configs[4] at its full size (4096 channels @10 MS/s, D = 46, 4096-tap IF FIR): the library's layout
    for long filters (heavy stream) and the two it does not choose, 150 calls each."""
    import torch
    pkg = load_package()
    fs, D, order, C, calls = 10e6, 46, 4096, 4096, 150 * SCALE
    params = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
    gen = fmsig.DeviceGenerator([fmsig.channel_params(fs, c % 2048) for c in range(C)], "cuda")
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(RING)]
    for k in range(RING):
        gen.generate(iq[k], k * N, N)
    w = _weights(torch, C, _a_stride(pkg, params))
    make = lambda: pkg.Batch(params, C, record_callbacks=False)
    base = _run(torch, make, iq, N, w, concurrency=0, lag=0, calls=calls, keep_audio=3)
    picks = [0, 63, C // 2, C - 1]
    _against_oracle(base, [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
                           for _ in picks],
                    lambda i: [iq[i % RING][c].cpu().numpy().reshape(-1) for c in picks])
    for name, kw in OVERLAPPED.items():
        _same(name, _run(torch, make, iq, N, w, calls=calls, **kw), base)


def test_config3_scaled_out_every_schedule_the_same_bits(oracle, fmsig):
    """BASELINE configs[2] scaled out to a full shard: 32 captures x 256 channels (cFineTuner table_size 256,
    FmDecode.h:42) in one batch, one input row per capture, 200 calls a schedule."""
    import torch
    pkg = load_package()
    fs, D, G, k, T, calls = 2.4e6, 11, 32, 256, 256, 200 * SCALE
    C = G * k
    offs = (-600e3, -360e3, -150e3, 75e3, 300e3, 600e3)
    iq = [torch.empty((G, N, 2), dtype=torch.float32, device="cuda") for _ in range(RING)]
    tmp = torch.empty((len(offs), N, 2), dtype=torch.float32, device="cuda")
    for g in range(G):
        gen = fmsig.DeviceGenerator(
            [fmsig.default_params(fs, f_offset=f0, amp=0.12, noise_sigma=0.004, seed=50 + i + 16 * g,
                                  pi=0x5000 + i + 16 * g, ps="CAP%05d" % (i + 16 * g), f_left=500.0 + 300 * i + 7 * g)
             for i, f0 in enumerate(offs)], "cuda")
        for r in range(RING):
            gen.generate(tmp, r * N, N)
            iq[r][g] = tmp.sum(dim=0)
    shifts = (np.arange(C, dtype=np.int32) % T) - T // 2
    params = pkg.make_params(fs, 0.0, 48000.0, 15000.0, D, table_size=T)
    w = _weights(torch, C, _a_stride(pkg, params))

    def make():
        b = pkg.Batch(params, C, tuning_shifts=shifts, record_callbacks=False)
        b.set_channels_per_capture(k)
        return b

    base = _run(torch, make, iq, N, w, concurrency=0, lag=0, calls=calls, keep_audio=4)
    assert len(base[1]) > 1000
    picks = [0, 63, C // 2, C - 1]
    _against_oracle(base, [oracle.OracleDecoder(fs, 0.0, 48000.0, 15000.0, D, table_size=T,
                                                tuning_shift=int(shifts[c])) for c in picks],
                    lambda i: [iq[i % RING][c // k].cpu().numpy().reshape(-1) for c in picks])
    for name, kw in OVERLAPPED.items():
        _same(name, _run(torch, make, iq, N, w, calls=calls, **kw), base)
