#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run in the build container).

The reference has no fixtures of its own and cannot be built here, so these vectors are
OUTPUTS OF THE ORACLE (which is pinned against the reference's recorded known answers, see
oracle/fmd_oracle.h); they pin (1) the oracle against drift of compiler / libm on another
host and (2) the HIP path against the oracle without needing the oracle at test time.

    python tools/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402
from tools import fmsig_py  # noqa: E402

N = 65536
CASES = {
    "stereo_rds_2p4M": dict(fs=2.4e6, D=11, nblk=48, noise=0.01, keep=(0, 1, 20, 47)),
    "stereo_rds_1p0M": dict(fs=1.0e6, D=4, nblk=20, noise=0.01, keep=(0, 1, 19)),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make(name, fs, D, nblk, noise, keep):
    p = fmsig_py.default_params(fs, noise_sigma=noise, seed=7)
    dec = oracle_py.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    out = {"fs": fs, "D": D, "nblk": nblk, "noise": noise, "seed": 7,
           "if_taps": dec.if_taps(), "audio_taps": dec.audio_taps(), "rds_lpf_taps": dec.rds_lpf_taps(),
           "rds_mf_taps": dec.rds_mf_taps(), "lut": dec.lut().view(np.float32)}
    iq_hash, audio_hash, counts, status = [], [], [], []
    for b in range(nblk):
        u8 = fmsig_py.generate_u8(p, b * N, N)
        iq_hash.append(sha(u8))
        audio = dec.process_stream(fmsig_py.u8_to_f32(u8))
        audio_hash.append(sha(audio))
        counts.append(audio.size)
        st = dec.status()
        status.append([st.stereo, st.tuning_offset, st.if_level, st.baseband_level, st.pilot_level,
                       st.rds_state])
        if b in keep:
            out["audio_%d" % b] = audio
            t = dec.taps()
            out["demod_head_%d" % b] = t["demod"][:256].view(np.float32)
            out["baseband_head_%d" % b] = t["baseband"][:256]
    out["iq_sha256"] = np.array(iq_hash)
    out["audio_sha256"] = np.array(audio_hash)
    out["audio_counts"] = np.array(counts, dtype=np.int32)
    out["status"] = np.array(status, dtype=np.float32)
    g = dec.rds_groups()
    out["rds_groups"] = np.array([[ci, *blk] for ci, blk in g], dtype=np.int32).reshape(-1, 5)
    frames = dec.uecp_frames()
    out["uecp_frames"] = np.array([f.hex() for f in frames])
    out["channel_name"] = np.array(dec.channel_name())
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes,", len(g), "groups,", len(frames), "frames")


# Sequences of calls of very different sizes: the short-block regimes of the half-band chain (fewer
# than L / than 2 (L - 1) inputs per stage), blocks shorter than a filter's history, the 11-tap first
# stage (baseband rate >= 320 kHz), a long IF filter.  One sha256 per call.
RAGGED = {
    "ragged_2p4M": dict(fs=2.4e6, D=11, order=0,
                        sizes=[65536, 88, 89, 100, 150, 170, 171, 200, 330, 500, 1000, 1900, 65536, 3000, 3662,
                               3663, 160, 5000, 88, 97, 8191, 640, 2222, 65536, 310, 320, 460, 470, 930, 940]),
    "ragged_hb11_400k": dict(fs=400e3, D=1, order=0,
                             sizes=[32000, 20, 21, 64, 500, 33, 2000, 32001, 25, 20000, 8193]),
    "ragged_longfir_10M": dict(fs=10e6, D=46, order=4096, sizes=[65536, 2000, 3000, 4095, 4097, 1000, 40000, 700]),
}


def make_ragged(name, fs, D, order, sizes):
    p = fmsig_py.default_params(fs, noise_sigma=0.01, seed=13)
    dec = oracle_py.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
    iq_hash, audio_hash, counts = [], [], []
    start = 0
    for n in sizes:
        u8 = fmsig_py.generate_u8(p, start, n)
        start += n
        iq_hash.append(sha(u8))
        audio = dec.process_stream(fmsig_py.u8_to_f32(u8))
        audio_hash.append(sha(audio))
        counts.append(audio.size)
    st = dec.status()
    out = {"fs": fs, "D": D, "order": order, "seed": 13, "noise": 0.01, "sizes": np.array(sizes, dtype=np.int32),
           "iq_sha256": np.array(iq_hash), "audio_sha256": np.array(audio_hash),
           "audio_counts": np.array(counts, dtype=np.int32),
           "rds_hb_lengths": np.array(dec.rds_hb_lengths(), dtype=np.int32),
           "status": np.array([st.stereo, st.tuning_offset, st.if_level, st.baseband_level, st.pilot_level,
                               st.rds_state], dtype=np.float32)}
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes,", len(sizes), "calls")


def receiver_session(rx, fmsig, p, nblk, u8=False, ahead=2):
    """One demux session written like the reference's two threads run: the source keeps `ahead`
    blocks queued, the demuxer pulls packets until the source ends.  Returns the packet list and
    the signal status sampled after every audio packet.  Shared by the generator and the tests."""
    packets, status = [], []
    written = 0

    def feed():
        nonlocal written
        while written < nblk and rx.queued_samples() < ahead * N:
            buf = fmsig.generate_u8(p, written * N, N)
            if u8:
                rx.write_u8(buf)
            else:
                rx.write_iq(fmsig.u8_to_f32(buf))
            written += 1
        if written == nblk:
            rx.end()

    while True:
        feed()
        pkt = rx.demux_read()
        if pkt is None:
            break
        packets.append(pkt)
        if pkt[0] == 1:
            status.append((rx.signal_status(), rx.pvr_signal_status()))
    return packets, status


def make_receiver(name="receiver_2p4M", fs=2.4e6, D=11, nblk=40):
    p = fmsig_py.default_params(fs, noise_sigma=0.01, seed=11, ps=" GOLD FM")
    rx = oracle_py.OracleReceiver(fs, -0.15 * fs, D, tuner_freq=99.9e6 + 0.15 * fs)
    assert rx.signal_status() is None  # stream change pending (RadioReceiver.cpp:548)
    packets, status = receiver_session(rx, fmsig_py, p, nblk)
    out = {"fs": fs, "D": D, "nblk": nblk, "seed": 11,
           "stream_id": np.array([k[0] for k in packets], dtype=np.int32),
           "pts": np.array([k[1] for k in packets], dtype=np.float64),
           "duration": np.array([k[2] for k in packets], dtype=np.float64),
           "size": np.array([len(k[3]) for k in packets], dtype=np.int32),
           "data_sha256": np.array([hashlib.sha256(k[3]).hexdigest() for k in packets]),
           "rds_payload": np.array([k[3].hex() for k in packets if k[0] == 2]),
           "signal": np.array([[s[0][0], s[0][1], float(s[0][2])] for s in status], dtype=np.float32),
           "pvr_signal_snr": np.array([[s[1]["signal"], s[1]["snr"]] for s in status], dtype=np.int64),
           "pvr_status_text": np.array([s[1]["adapter_status"] for s in status]),
           "provider_name": np.array(status[-1][1]["provider_name"]),
           "audio_level": np.array(rx.audio_level(), dtype=np.float32)}
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes,", len(packets), "packets,",
          int((out["stream_id"] == 2).sum()), "rds packets")


# design-level vectors.  (order, cutoff) pairs, filters and tables the path builds.
G1_LANCZOS = [(32, 0.15), (88, 0.6 / 11), (250, 0.06), (218, 15000 / 218181.8), (368, 0.6 / 46),
              (4096, 0.6 / 46)]
G1_KAISER = {  # name: (scale, astop, fpass, fstop, fs) -- the calls of FmDecode.cpp:286 (29 taps)
    # and RDSProcess.cpp:97 (75 taps) at the two RDS process rates the half-band chains end at
    "audio_lpf_48k": (1.0, 60.0, 15000.0, float(np.float32(1.4 * 15000.0)), 48000.0),
    "rds_lpf_27272": (1.0, 40.0, 2400.0, float(np.float32(1.3 * 2400.0)), float(np.float32(2.4e6 / 11 / 8))),
    "rds_lpf_31250": (1.0, 40.0, 2400.0, float(np.float32(1.3 * 2400.0)), 31250.0),
}
G1_BIQUAD = {  # name: (type, f0, q, fs): 19 kHz notch (FmDecode.cpp:291), bit-sync resonator
    "notch_19k_48k": (3, 19000.0, 5.0, 48000.0),
    "bitsync_27272": (2, 1187.5, 500.0, float(np.float32(2.4e6 / 11 / 8))),
    "bitsync_31250": (2, 1187.5, 500.0, 31250.0),
}
G1_LUT = [(64, 10), (64, -7), (64, 0), (64, 31), (256, -128), (256, 99)]


def design_vectors(mod):
    """The G1 set computed by `mod` (oracle_py or the product package: same function names)."""
    out = {}
    for order, cutoff in G1_LANCZOS:
        out["lanczos_%d_%.6f" % (order, cutoff)] = mod.design_lanczos(order, cutoff)
    for name, a in G1_KAISER.items():
        out["kaiser_" + name] = mod.design_lp_kaiser(*a)
    for name, a in G1_BIQUAD.items():
        out["biquad_" + name] = mod.design_biquad(*a)
    for size, shift in G1_LUT:
        out["lut_%d_%d" % (size, shift)] = mod.design_tuner_lut(size, shift)
    return out


def make_design(name="design_g1"):
    out = design_vectors(oracle_py)
    for fs, D in ((2.4e6, 11), (1.0e6, 4)):
        dec = oracle_py.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
        out["rds_mf_%d" % int(fs / D)] = dec.rds_mf_taps()
        out["rds_hb_%d" % int(fs / D)] = np.array(dec.rds_hb_lengths(), dtype=np.int32)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    for name, kw in CASES.items():
        make(name, **kw)
    for name, kw in RAGGED.items():
        make_ragged(name, **kw)
    make_receiver()
    make_design()
