"""GPU: the stream side of cRadioReceiver on the HIP decoder against the
oracle's restatement and the committed golden session: every demux packet (stream id, PTS,
duration, payload bytes), the audio level meter computed on the device, both GetSignalStatus."""
import hashlib
import os
import threading

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package
from tools.make_golden import receiver_session

pytestmark = pytest.mark.gpu
N = 65536


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def _f32bits(x):
    return np.float32(x).view(np.uint32)


def test_golden_session(pkg, fmsig):
    g = np.load(os.path.join(ROOT, "tests", "golden", "receiver_2p4M.npz"))
    fs, D, nblk = float(g["fs"]), int(g["D"]), int(g["nblk"])
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=int(g["seed"]), ps=" GOLD FM")
    rx = pkg.Receiver(fs, -0.15 * fs, D, tuner_freq=99.9e6 + 0.15 * fs)
    assert rx.signal_status() is None
    packets, status = receiver_session(rx, fmsig, p, nblk)
    assert [k[0] for k in packets] == list(g["stream_id"])
    assert np.array_equal(np.array([k[1] for k in packets]), g["pts"])
    assert np.array_equal(np.array([k[2] for k in packets]), g["duration"])
    assert [hashlib.sha256(k[3]).hexdigest() for k in packets] == list(g["data_sha256"])
    sig = np.array([[s[0][0], s[0][1], float(s[0][2])] for s in status], dtype=np.float32)
    assert np.array_equal(sig.view(np.uint32), g["signal"].view(np.uint32))
    assert [s[1]["adapter_status"] for s in status] == list(g["pvr_status_text"])
    assert [[s[1]["signal"], s[1]["snr"]] for s in status] == g["pvr_signal_snr"].tolist()
    assert status[-1][1]["provider_name"] == "GOLD FM"
    rx.close()


@pytest.mark.parametrize("u8", [False, True])
def test_packets_and_status_equal_oracle(pkg, oracle, fmsig, u8):
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.02, seed=3, pi=0xFDFE, ps="ESC \xff\xfd ")
    o = oracle.OracleReceiver(fs, -0.15 * fs, D, tuner_freq=101.3e6, adapter_name="rtl #1")
    r = pkg.Receiver(fs, -0.15 * fs, D, tuner_freq=101.3e6, adapter_name="rtl #1")
    po, so = receiver_session(o, fmsig, p, 50, u8=u8)
    pr, sr = receiver_session(r, fmsig, p, 50, u8=u8)
    assert len(po) == len(pr)
    for a, b in zip(po, pr):
        assert a == b  # stream id, pts, duration (doubles), payload bytes
    assert sum(k[0] == 2 for k in pr) >= 2
    for (a3, apvr), (b3, bpvr) in zip(so, sr):
        assert _f32bits(a3[0]) == _f32bits(b3[0]) and _f32bits(a3[1]) == _f32bits(b3[1])
        assert a3[2] == b3[2]
        assert apvr == bpvr
    assert r.demux_read() is None and o.demux_read() is None
    r.close()


def test_audio_level_meter_on_device(pkg, oracle, fmsig):
    """m_AudioLevel / SamplesMeanRMS (RadioReceiver.cpp:526-528, 584-598) computed in the audio
    kernel: batch of channels against one oracle receiver each."""
    fs, D, Cn = 2.4e6, 11, 5
    ps = [fmsig.channel_params(fs, c) for c in range(Cn)]
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), Cn)
    os_ = [oracle.OracleReceiver(fs, -0.15 * fs, D) for _ in range(Cn)]
    for o in os_:
        assert o.demux_read()[0] == -11
    for blk in range(8):
        iq = np.stack([fmsig.generate_f32(ps[c], blk * N, N) for c in range(Cn)])
        b.process_host(iq.view(np.complex64))
        for c in range(Cn):
            os_[c].write_iq(iq[c])
            k = os_[c].demux_read()
            while k[0] != 1:
                k = os_[c].demux_read()
            mo, ro, lo = os_[c].audio_level()
            mg, rg, lg = b.audio_level(c)
            assert (_f32bits(mo), _f32bits(ro), _f32bits(lo)) == (_f32bits(mg), _f32bits(rg), _f32bits(lg))
    b.reset()  # cFmDecoder::Reset does not touch the receiver's meter
    assert _f32bits(b.audio_level(0)[2]) == _f32bits(os_[0].audio_level()[2])
    b.close()


def test_source_thread_and_demux_thread(pkg, oracle, fmsig):
    """The reference's two threads: a source thread calling WriteDataBuffer / EndDataBuffer and
    the demuxer blocking in DemuxRead (SourceGetSamples waits, RadioReceiver.cpp:445-460)."""
    fs, D, nblk = 2.4e6, 11, 12
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=21)
    blocks = [fmsig.generate_f32(p, k * N, N) for k in range(nblk)]
    r = pkg.Receiver(fs, -0.15 * fs, D)
    o = oracle.OracleReceiver(fs, -0.15 * fs, D)

    def source():
        import time
        for blk in blocks:
            time.sleep(0.03)
            r.write_iq(blk)
        r.end()

    t = threading.Thread(target=source)
    t.start()
    got = []
    while True:
        k = r.demux_read()  # blocks while the queue is empty and the end is not marked
        if k is None:
            break
        got.append(k)
    t.join()
    for blk in blocks:
        o.write_iq(blk)
    o.end()
    exp = []
    while True:
        k = o.demux_read()
        if k is None:
            break
        exp.append(k)
    assert got == exp
    assert r.queued_samples() == 0
    r.close()
