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


if __name__ == "__main__":
    for name, kw in CASES.items():
        make(name, **kw)
