"""Golden vectors (tests/golden/*.npz, made by tools/make_golden.py from the oracle).

CPU: the oracle built on THIS host still reproduces them (compiler / libm drift would show).
GPU: the HIP path reproduces them bit for bit, without the oracle in the loop."""
import glob
import hashlib
import os

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

N = 65536
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stereo_rds_*.npz")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _params(g, fmsig):
    return fmsig.default_params(float(g["fs"]), noise_sigma=float(g["noise"]), seed=int(g["seed"]))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path, oracle, fmsig):
    g = np.load(path)
    fs, D, nblk = float(g["fs"]), int(g["D"]), int(g["nblk"])
    p = _params(g, fmsig)
    dec = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    assert np.array_equal(dec.if_taps().view(np.uint32), g["if_taps"].view(np.uint32))
    assert np.array_equal(dec.audio_taps().view(np.uint32), g["audio_taps"].view(np.uint32))
    assert np.array_equal(dec.rds_lpf_taps().view(np.uint32), g["rds_lpf_taps"].view(np.uint32))
    assert np.array_equal(dec.rds_mf_taps().view(np.uint32), g["rds_mf_taps"].view(np.uint32))
    for b in range(nblk):
        u8 = fmsig.generate_u8(p, b * N, N)
        assert sha(u8) == str(g["iq_sha256"][b]), "generator drift in block %d" % b
        audio = dec.process_stream(fmsig.u8_to_f32(u8))
        assert sha(audio) == str(g["audio_sha256"][b]), "oracle drift in block %d" % b
    assert [f.hex() for f in dec.uecp_frames()] == [str(x) for x in g["uecp_frames"]]
    assert dec.channel_name() == str(g["channel_name"])
    got = np.array([[ci, *blk] for ci, blk in dec.rds_groups()], dtype=np.int32).reshape(-1, 5)
    assert np.array_equal(got, g["rds_groups"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_reproduces_golden(path, fmsig):
    pkg = load_package()
    g = np.load(path)
    fs, D, nblk = float(g["fs"]), int(g["D"]), int(g["nblk"])
    p = _params(g, fmsig)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    assert np.array_equal(b.design("if_taps").view(np.uint32), g["if_taps"].view(np.uint32))
    for blk in range(nblk):
        iq = fmsig.u8_to_f32(fmsig.generate_u8(p, blk * N, N))
        audio = b.process_host(iq.view(np.complex64), shared=True)[0]
        assert audio.size == int(g["audio_counts"][blk])
        key = "audio_%d" % blk
        if key in g:
            rms = float(np.sqrt(np.mean((audio.astype(np.float64) - g[key]) ** 2)))
            assert rms <= 1e-5, (blk, rms)  # the contract (BASELINE north_star)
            assert np.array_equal(b.tap("demod")[:256].view(np.float32), g["demod_head_%d" % blk])
        assert sha(audio) == str(g["audio_sha256"][blk]), "block %d differs from the oracle" % blk
        st = b.status()
        ref = g["status"][blk]
        assert st.stereo_detected == int(ref[0]) and st.rds_state == int(ref[5])
        assert np.float32(st.pilot_level) == ref[4] and np.float32(st.interface_level) == ref[2]
    assert [f.hex() for f in b.sink.frames.get(0, [])] == [str(x) for x in g["uecp_frames"]]
    assert b.sink.names.get(0, "") == str(g["channel_name"])


RAGGED = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ragged_*.npz")))


def _ragged_calls(g, fmsig):
    p = fmsig.default_params(float(g["fs"]), noise_sigma=float(g["noise"]), seed=int(g["seed"]))
    start = 0
    for k, n in enumerate(int(x) for x in g["sizes"]):
        u8 = fmsig.generate_u8(p, start, n)
        start += n
        assert sha(u8) == str(g["iq_sha256"][k]), "generator drift in call %d" % k
        yield k, n, fmsig.u8_to_f32(u8)


@pytest.mark.parametrize("path", RAGGED, ids=[os.path.basename(f) for f in RAGGED])
def test_oracle_reproduces_ragged_golden(path, oracle, fmsig):
    """Call sequences through the short-block regimes (half-band stages below L and below 2 (L - 1)
    inputs, blocks shorter than a filter's history), the 11-tap first stage and a 4096-tap IF filter."""
    g = np.load(path)
    fs, D, order = float(g["fs"]), int(g["D"]), int(g["order"])
    dec = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order)
    assert dec.rds_hb_lengths() == [int(x) for x in g["rds_hb_lengths"]]
    for k, n, iq in _ragged_calls(g, fmsig):
        audio = dec.process_stream(iq)
        assert audio.size == int(g["audio_counts"][k]), (k, n)
        assert sha(audio) == str(g["audio_sha256"][k]), "oracle drift in call %d (%d samples)" % (k, n)


@pytest.mark.gpu
@pytest.mark.parametrize("path", RAGGED, ids=[os.path.basename(f) for f in RAGGED])
def test_hip_path_reproduces_ragged_golden(path, fmsig):
    pkg = load_package()
    g = np.load(path)
    fs, D, order = float(g["fs"]), int(g["D"]), int(g["order"])
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D, if_filter_order=order), 1)
    for k, n, iq in _ragged_calls(g, fmsig):
        audio = b.process_host(iq.view(np.complex64), shared=True)[0]
        assert audio.size == int(g["audio_counts"][k]), (k, n)
        assert sha(audio) == str(g["audio_sha256"][k]), "call %d (%d samples) differs from the oracle" % (k, n)
    st, ref = b.status(), g["status"]
    assert st.stereo_detected == int(ref[0]) and st.rds_state == int(ref[5])
    assert np.float32(st.pilot_level) == ref[4] and np.float32(st.interface_level) == ref[2]
    b.close()
