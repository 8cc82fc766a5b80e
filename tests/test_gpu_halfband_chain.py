"""The three half-band stages of the RDS decimation chain as one stream with the intermediate rows in
LDS (k_halfband_chain, what large batches run) against the CPU oracle and against one launch per stage
(k_halfband4), bit for bit.  The library picks the fused form by batch size; here it is forced
(fmd_batch_debug_set "halfband_chain"), with as many stretches as the call allows, and alternated
with the per-stage form from call to call (both keep the stages' delay lines in the same place).
Reference: CHalfBandDecimateBy2::DecBy2, /root/reference/src/DownConvert.cpp:512-550."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
N = 65536


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("fs,D,sizes,alternate,nomix", [
    (2.4e6, 11, [N] * 6 + [40000, 3000, 65535, 2000, 50001, N], False, 1),  # 15 / 23 / 43 taps
    (2.4e6, 11, [N, 30000, N, N, 1500, N, 65533, N], True, 1),              # fused and per-stage calls in turn
    (2.4e6, 11, [N, 30000, N, N, 1500, N, 65533, N], True, 0),              # ... with the serial stage writing the mixed rows
    (1.0e6, 4, [N] * 4 + [30000, 1200, 65535, N], False, 1),                # 15 / 19 / 35 taps
    (10e6, 46, [N] * 4, False, 1),
])
def test_fused_chain_rds_taps_bit_exact(oracle, fmsig, fs, D, sizes, alternate, nomix):
    """nomix = 1 (what large batches run): the serial stage writes no mixed rows, the chain multiplies the
    baseband with the batch-wide oscillator sequence (computed by the host, once per call) itself; calls that take a launch per stage
    in between (the alternating case, the 1500- and 3000-sample calls) read mixed rows again and must find
    the rows of history and the per-channel oscillator state the other form left for them."""
    pkg = load_package()
    p = fmsig.default_params(fs, noise_sigma=0.01, seed=41)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D)
    b = pkg.Batch(pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D), 1)
    b.debug_set("halfband_chain", 1)  # before the first call: starts the oscillator sequence too
    b.debug_set("nomix", nomix)
    b.enable_taps()
    pos = 0
    for blk, n in enumerate(sizes):
        b.debug_set("halfband_chain", 0 if (alternate and blk % 2) else 1)
        iq = fmsig.generate_f32(p, pos, n)
        pos += n
        a_ref = o.process_stream(iq)
        a_gpu = b.process_host(iq.view(np.complex64), shared=True)[0]
        taps = o.taps()
        for name in ("rds_lpf", "rds_pll", "rds_mf"):
            g, r = b.tap(name), taps[name]
            assert g.shape == r.shape, (blk, name, g.shape, r.shape)
            assert _bits_equal(g.view(np.float32), r.view(np.float32)), (blk, n, name)
        assert _bits_equal(a_gpu, a_ref), (blk, n)
    assert b.sink.frames.get(0, []) == o.uecp_frames()
    b.close()


@pytest.mark.parametrize("C", [130, 256])
def test_fused_chain_equals_per_stage(oracle, fmsig, C):
    """Many channels (a ragged last group of lanes): the UECP frames and the status of both forms
    identical on every channel, a few channels also against the oracle."""
    pkg = load_package()
    fs, D = 2.4e6, 11
    base = [fmsig.default_params(fs, noise_sigma=0.01, seed=400 + k, pi=0x4000 + k, ps="CHAIN%03d" % k)
            for k in range(3)]
    par = pkg.make_params(fs, -0.15 * fs, 48000.0, 15000.0, D)
    fused, plain = pkg.Batch(par, C), pkg.Batch(par, C)
    fused.debug_set("halfband_chain", 1)
    plain.debug_set("halfband_chain", 0)
    refs = [oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, D) for _ in range(3)]
    for blk in range(30):
        src = [fmsig.generate_f32(p, blk * N, N) for p in base]
        iq = np.stack([src[c % 3] for c in range(C)]).view(np.complex64).reshape(C, N)
        a_f, a_p = fused.process_host(iq), plain.process_host(iq)
        assert _bits_equal(a_f, a_p), blk
        for k in range(3):
            refs[k].process_stream(src[k])
    assert fused.sink.frames == plain.sink.frames and len(fused.sink.frames) == C
    for c in (0, 1, 2, C - 1):
        assert fused.sink.frames[c] == refs[c % 3].uecp_frames(), c
        assert fused.status(c).rds_state == plain.status(c).rds_state == refs[c % 3].status().rds_state
    fused.close()
    plain.close()
