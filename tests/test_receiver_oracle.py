"""CPU: the oracle's restatement of cRadioReceiver's stream members (DemuxRead, PTS clock,
byte-stuffed RDS packets, signal status) against the committed golden
session and against the arithmetic the reference spells out."""
import hashlib
import os

import numpy as np

from __graft_entry__ import ROOT
from tools.make_golden import receiver_session

N = 65536


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "receiver_2p4M.npz"))


def test_oracle_receiver_matches_golden_session(oracle, fmsig):
    g = _golden()
    fs, D, nblk = float(g["fs"]), int(g["D"]), int(g["nblk"])
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=int(g["seed"]), ps=" GOLD FM")
    rx = oracle.OracleReceiver(fs, -0.15 * fs, D, tuner_freq=99.9e6 + 0.15 * fs)
    packets, status = receiver_session(rx, fmsig, p, nblk)
    assert [k[0] for k in packets] == list(g["stream_id"])
    assert np.array_equal(np.array([k[1] for k in packets]), g["pts"])
    assert np.array_equal(np.array([k[2] for k in packets]), g["duration"])
    assert [hashlib.sha256(k[3]).hexdigest() for k in packets] == list(g["data_sha256"])
    sig = np.array([[s[0][0], s[0][1], float(s[0][2])] for s in status], dtype=np.float32)
    assert np.array_equal(sig.view(np.uint32), g["signal"].view(np.uint32))
    assert [s[1]["adapter_status"] for s in status] == list(g["pvr_status_text"])
    assert [[s[1]["signal"], s[1]["snr"]] for s in status] == g["pvr_signal_snr"].tolist()
    assert status[-1][1]["provider_name"] == str(g["provider_name"]) == "GOLD FM"  # Trim()


def test_packet_order_and_pts_clock(oracle, fmsig):
    """RadioReceiver.cpp:462-542: stream-change packet first, RDS bytes before the next audio
    packet with the PTS the next audio packet will carry, duration = floats * 1e6 / 2 / 48000,
    pts accumulates from STREAM_TIME_BASE."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.005)
    rx = oracle.OracleReceiver(fs, -0.15 * fs, D)
    assert rx.signal_status() is None and rx.pvr_signal_status() is None  # :548, :563
    packets, _ = receiver_session(rx, fmsig, p, 45)
    assert packets[0] == (-11, 0.0, 0.0, b"")
    pts = 1000000.0
    seen_rds = 0
    for sid, ppts, dur, data in packets[1:]:
        assert ppts == pts
        if sid == 1:
            nfloats = len(data) // 4
            assert dur == float(nfloats) * 1000000 / 2 / 48000
            pts = pts + dur
        else:
            assert sid == 2 and dur == 0.0
            assert data[0] == 0xFE and data[-1] == 0xFF  # :397, :411
            seen_rds += 1
    assert seen_rds >= 2
    assert rx.demux_read() is None  # end marked, queue empty: nullptr (:447-459)
    rx.set_stream_change()
    assert rx.signal_status() is None
    assert rx.demux_read()[0] == -11
    assert rx.signal_status() is not None


def test_signal_status_arithmetic(oracle, fmsig):
    """:544-582 on the oracle's own getters: float log10, + 3.01 in double, int truncation."""
    fs, D = 2.4e6, 11
    p = fmsig.default_params(fs, noise_sigma=0.005)
    rx = oracle.OracleReceiver(fs, -0.15 * fs, D, tuner_freq=100.0e6)
    receiver_session(rx, fmsig, p, 24)  # pilot lock needs ~20 blocks
    if_db, au_db, stereo = rx.signal_status()
    mean, rms, level = rx.audio_level()
    lib = oracle.lib()
    import ctypes as C
    st = oracle.FmoStatus()
    lib.fmo_receiver_decoder.restype = C.c_void_p
    lib.fmo_receiver_decoder.argtypes = [C.c_void_p]
    lib.fmo_get_status(C.c_void_p(lib.fmo_receiver_decoder(rx._h)), C.byref(st))
    f32 = np.float32
    libm = C.CDLL("libm.so.6")  # the reference's log10(float) is glibc's log10f
    libm.log10f.restype = C.c_float
    libm.log10f.argtypes = [C.c_float]
    assert f32(if_db) == f32(20) * f32(libm.log10f(st.if_level))
    assert f32(au_db) == f32(np.float64(f32(20) * f32(libm.log10f(level))) + 3.01)
    assert stereo == bool(st.stereo)
    pvr = rx.pvr_signal_status()
    assert pvr["signal"] == int(2.5 * (np.float64(f32(if_db)) + 40) * 656)
    assert pvr["snr"] == int(f32(f32(au_db) + f32(100)) * f32(656))
    # five conversions for six arguments: "IF=" carries the tuned frequency in MHz
    assert pvr["adapter_status"].startswith("Freq.=100.0000MHz - %s - IF=+99."
                                            % ("Stereo" if stereo else "Mono"))
    assert pvr["adapter_name"] == "Generic RTL2832U"


def test_uecp_buffer_limit_and_stuffing(oracle):
    """AddUECPDataFrame (:387-414): 0xFD..0xFF escaped as 0xFD (v & 3) - 1; frames are refused
    once more than 16384 bytes are pending."""
    import ctypes as C
    lib = oracle.lib()
    rx = oracle.OracleReceiver(2.4e6, -0.36e6, 11)
    assert rx.demux_read()[0] == -11
    lib.fmo_debug_push_group.argtypes = [C.c_void_p, C.POINTER(C.c_uint16)]
    lib.fmo_receiver_decoder.restype = C.c_void_p
    lib.fmo_receiver_decoder.argtypes = [C.c_void_p]
    dec = C.c_void_p(lib.fmo_receiver_decoder(rx._h))
    # type 0A groups with changing PI / PTY produce frames; PI 0xFDFE forces escapes
    for k in range(4):
        blocks = (C.c_uint16 * 4)(0xFDFE, (0 << 12) | (k & 3) | ((5 + k) << 5), 0xE0E0,
                                  (0x41 + k) << 8 | 0x42)
        lib.fmo_debug_push_group(dec, blocks)
    rx.write_iq(np.zeros(2 * 8192, dtype=np.float32))
    rx.end()
    pkts = []
    while True:
        k = rx.demux_read()
        if k is None:
            break
        pkts.append(k)
    rds = [k for k in pkts if k[0] == 2]
    assert len(rds) == 1
    data = rds[0][3]
    assert b"\xfd\x01\xfd\x00" in data  # PI bytes 0xFE, 0xFD -> FD 01, FD 00
    assert data.count(b"\xfe") == data.count(b"\xff")  # only frame delimiters remain unescaped
    # more than 16384 pending bytes: further frames are refused until DemuxRead drains the buffer
    for k in range(3000):
        blocks = (C.c_uint16 * 4)(0x1234, (0 << 12) | (k & 3) | ((k % 31) << 5), 0xE0E0, 0x2020)
        lib.fmo_debug_push_group(dec, blocks)
    rx2 = oracle.OracleReceiver(2.4e6, -0.36e6, 11)
    assert rx2.demux_read()[0] == -11
    dec2 = C.c_void_p(lib.fmo_receiver_decoder(rx2._h))
    for k in range(3000):
        blocks = (C.c_uint16 * 4)(0x1234, (0 << 12) | (k & 3) | ((k % 31) << 5), 0xE0E0, 0x2020)
        lib.fmo_debug_push_group(dec2, blocks)
    rx2.write_iq(np.zeros(2 * 8192, dtype=np.float32))
    rx2.end()
    assert rx2.demux_read()[0] == 1
    big = rx2.demux_read()
    assert big[0] == 2 and 16384 < len(big[3]) < 16384 + 64
    assert rx2.demux_read() is None
