"""The getters of cFmDecoder from a second thread, and what happens when RDS groups are not drained.

Kodi's status thread calls StereoDetected / GetInterfaceLevel / ... while the demux thread is inside
ProcessStream (/root/reference/src/RadioReceiver.cpp:544-572 against :524, no common lock).  Here the
getters read a per-channel record that the last kernel of every call writes into host-mapped
memory (include/fmd.h, fmd_batch_get_status): no device call, nothing of the decoder is touched."""
import threading

import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def test_getters_polled_from_a_second_thread(oracle, fmsig):
    """One thread runs ProcessStream for 100 blocks, another polls all five getters as fast as it can
    (ctypes releases the GIL inside the library: the calls really overlap).  The audio stays bit-identical
    to the oracle's, and every value a poll returns is a value the decoder had after some completed
    call (or the fresh decoder's zero)."""
    pkg = load_package()
    fs, D, nblk = 2.4e6, 11, 100
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=33)
    blocks = [fmsig.generate_f32(p, k * N, N) for k in range(nblk)]
    ref = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    seen = {"stereo": {0}, "tuning": {_bits(ref.status().tuning_offset)}, "if": {_bits(0.0)},
            "bb": {_bits(0.0)}, "pilot": {_bits(0.0)}}
    want_audio = []
    for k in range(nblk):
        want_audio.append(ref.process_stream(blocks[k]))
        s = ref.status()
        seen["stereo"].add(int(s.stereo))
        seen["tuning"].add(_bits(s.tuning_offset))
        seen["if"].add(_bits(s.if_level))
        seen["bb"].add(_bits(s.baseband_level))
        seen["pilot"].add(_bits(s.pilot_level))

    dec = pkg.FmDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    polls = {k: [] for k in seen}
    stop = threading.Event()
    errors = []

    def poll():
        try:
            while not stop.is_set():
                polls["stereo"].append(int(dec.StereoDetected()))
                polls["tuning"].append(_bits(dec.GetTuningOffset()))
                polls["if"].append(_bits(dec.GetInterfaceLevel()))
                polls["bb"].append(_bits(dec.GetBasebandLevel()))
                polls["pilot"].append(_bits(dec.GetPilotLevel()))
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    t = threading.Thread(target=poll)
    t.start()
    got_audio = [dec.ProcessStream(blocks[k].view(np.complex64)) for k in range(nblk)]
    stop.set()
    t.join(timeout=30)
    assert not errors, errors
    for k in range(nblk):
        assert np.array_equal(got_audio[k].view(np.uint32), want_audio[k].view(np.uint32)), k
    for name, vals in polls.items():
        assert len(vals) > nblk, (name, len(vals))  # the poller really ran beside the calls
        stray = set(vals) - seen[name]
        assert not stray, (name, sorted(stray)[:4])
    assert len(set(polls["if"])) > nblk // 4  # and it saw the decoder's state move
    # after the last call the getters hold exactly the final status
    s = ref.status()
    assert (int(dec.StereoDetected()), _bits(dec.GetInterfaceLevel()), _bits(dec.GetBasebandLevel()),
            _bits(dec.GetPilotLevel()), _bits(dec.GetTuningOffset())) == (
        int(s.stereo), _bits(s.if_level), _bits(s.baseband_level), _bits(s.pilot_level),
        _bits(s.tuning_offset))
    dec.Reset()
    ref.reset()
    s = ref.status()
    assert (int(dec.StereoDetected()), _bits(dec.GetInterfaceLevel()), _bits(dec.GetBasebandLevel()),
            _bits(dec.GetPilotLevel())) == (int(s.stereo), _bits(s.if_level), _bits(s.baseband_level),
                                            _bits(s.pilot_level))
    dec.close()


def test_status_is_one_calls_record_with_overlapped_calls(oracle, fmsig):
    """Device path, calls overlapped: the getters never block and never submit anything; once the
    stream has been ordered behind call k and synchronised, the snapshot is call k's or newer, and the
    index that comes with it says which."""
    import torch
    pkg = load_package()
    fs, D, C, nblk = 2.4e6, 11, 1024, 6
    chans = [fmsig.channel_params(fs, c % 8) for c in range(C)]
    gen = fmsig.DeviceGenerator(chans, torch.device("cuda"))
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    b.set_concurrency(2)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = [torch.empty((C, N, 2), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    audio = [torch.zeros((C, a_stride), dtype=torch.float32, device="cuda") for _ in range(nblk)]
    for k in range(nblk):
        gen.generate(iq[k], k * N, N)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    assert b.status_call_index(0) == 0 and b.status(5).interface_level == 0.0
    for k in range(nblk):
        b.process_device(iq[k].data_ptr(), N, N, audio[k].data_ptr(), a_stride, st)
        assert b.status_call_index(0) <= k + 1  # a poll in between: whatever is complete, no waiting
    b.wait(stream=st)
    torch.cuda.synchronize()
    assert b.status_call_index(0) == nblk and b.status_call_index(C - 1) == nblk
    ref = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    for k in range(nblk):
        ref.process_stream(iq[k][3].cpu().numpy().reshape(-1))
    so = ref.status()
    for c in (3, 11, C - 5):  # channels c % 8 == 3 carry the same station
        sg = b.status(c)
        assert (sg.stereo_detected, _bits(sg.interface_level), _bits(sg.baseband_level),
                _bits(sg.pilot_level), sg.rds_state) == (
            so.stereo, _bits(so.if_level), _bits(so.baseband_level), _bits(so.pilot_level), so.rds_state)
    b.close()


def test_lost_rds_groups_do_not_disable_the_batch(oracle, fmsig):
    """A record buffer that is too small for the groups queued loses the surplus and nothing else:
    the condition is reported once (FMD_WARN_RDS_LOST), the batch keeps taking calls and its audio stays
    bit-identical to the oracle's."""
    import torch
    pkg = load_package()
    fs, D, C, nblk = 2.4e6, 11, 64, 48
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=5)
    gen = fmsig.DeviceGenerator([p] * C, torch.device("cuda"))
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), C, record_callbacks=False)
    ref = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    a_stride = (b.max_audio_floats(N) + 63) // 64 * 64
    iq = torch.empty((C, N, 2), dtype=torch.float32, device="cuda")
    audio = torch.zeros((C, a_stride), dtype=torch.float32, device="cuda")
    rec = torch.zeros((2, 4), dtype=torch.int32, device="cuda")  # room for two groups only
    st = torch.cuda.current_stream().cuda_stream
    lost_reports = 0
    for k in range(nblk):
        gen.generate(iq, k * N, N)
        nf = b.process_device(iq.data_ptr(), N, N, audio.data_ptr(), a_stride, st)
        lost_reports += int(b.wait(stream=st))
        if k == 40:  # by now every channel has produced groups that nobody drained
            b.export_rds_device(rec.data_ptr(), 2, stream=st)
            torch.cuda.synchronize()
            assert int((rec[:, 0] != 0).sum()) == 2
        torch.cuda.synchronize()
        want = ref.process_stream(iq[7].cpu().numpy().reshape(-1))
        assert np.array_equal(audio[7, :nf].cpu().numpy().view(np.uint32), want.view(np.uint32)), k
    assert len(ref.rds_groups()) > 2  # there was more to lose than the buffer held
    assert lost_reports == 1 and not b.take_rds_lost()
    b.close()
