"""What the REFERENCE'S OWN CLASSES returned for inputs the whole-decoder streams of test_ref_streams.py cannot
reach (tests/golden/ref_groups.npz, ref_fir.npz, written by `python tools/ref_crosscheck.py --emit ...` in the
build container):

  * ref_groups.npz -- group lists pushed straight into cRDSGroupDecoder::DecodeRDS
    (/root/reference/src/RDSGroupDecoder.cpp:166-945; constructed in zeroed storage and Reset() like the
    member of a cFmDecoder, RDSProcess.cpp:80,94): every group type in both versions, open-data carriers, texts
    with missing segments, clocks, 6000 random groups of three stations, a station with PI 0.  Recorded: the
    UECP frames handed to AddUECPDataFrame and every name handed to SetChannelName.  Compared here with the
    product's host group decoder (csrc/fmd_groups.hpp through fmd_group_decoder_*) and with the oracle's.
  * ref_fir.npz -- cFineTuner + cDownsampleFilter(complex) on their own (FmDecode.cpp:45-82,
    DownConvert.cpp:63-154) at the parameters of BASELINE configs[2] (256-entry tuner table, one capture,
    eight tuning shifts) and configs[4] (4096 taps, D = 46) -- which the reference's cFmDecoder constructor
    never builds (FmDecode.cpp:249, 262) -- on full, ragged and shorter-than-the-filter calls.  Recorded: per
    call the output count and the SHA-256 of the FIR output.  Compared with the oracle's `demod` tap (CPU)
    and the HIP path's (`-m gpu`, fmd_batch_get_tap).

Data only; like ref_streams.npz it comes from a build with stand-in headers and pins nothing by the task's rules.
"""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

GROUPS = os.path.join(ROOT, "tests", "golden", "ref_groups.npz")
FIR = os.path.join(ROOT, "tests", "golden", "ref_fir.npz")


def _group_lists():
    z = np.load(GROUPS)
    n = len([k for k in z.files if k.endswith("_groups")])
    out = []
    for i in range(n):
        raw, at, frames = z["g%02d_frames" % i].tobytes(), 0, []
        for l in z["g%02d_frame_len" % i]:
            frames.append(raw[at:at + int(l)])
            at += int(l)
        names = z["g%02d_names" % i].tobytes()
        names = [names[k:k + 8].rstrip(b"\0").decode("latin1") for k in range(0, len(names), 8)]
        out.append((str(z["g%02d_name" % i]), z["g%02d_groups" % i], frames, names))
    return out


def _fir_cases():
    z = np.load(FIR)
    n = len([k for k in z.files if k.endswith("_def")])
    return [(json.loads(str(z["f%02d_def" % i])), z["f%02d_sha256" % i], z["f%02d_count" % i]) for i in range(n)]


@pytest.fixture(scope="module")
def pkg():
    return load_package()


GROUP_LISTS = _group_lists()
FIR_CASES = _fir_cases()


@pytest.mark.parametrize("rec", GROUP_LISTS, ids=[r[0].split(":")[0].replace(" ", "_")[:40] for r in GROUP_LISTS])
def test_product_group_decoder_reproduces_the_reference_frames(pkg, rec):
    name, groups, frames, names = rec
    assert len(frames) > 5
    gd = pkg.GroupDecoder()
    got_names = []
    for g in groups:
        before = gd.name
        gd.push([int(x) for x in g])
        if gd.name != before:
            got_names.append(gd.name)
    assert gd.frames == frames, name
    assert gd.name[:8] == (names[-1] if names else "")


@pytest.mark.parametrize("rec", GROUP_LISTS, ids=[r[0].split(":")[0].replace(" ", "_")[:40] for r in GROUP_LISTS])
def test_oracle_group_decoder_reproduces_the_reference_frames(oracle, rec):
    name, groups, frames, names = rec
    L = oracle.lib()
    L.fmo_debug_push_group.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    o = oracle.OracleDecoder(2.4e6, -0.36e6, 48000.0, 15000.0, 11)
    for g in groups:
        L.fmo_debug_push_group(o._h, (ctypes.c_uint16 * 4)(*[int(x) for x in g]))
    assert o.uecp_frames() == frames, name
    assert o.channel_name()[:8] == (names[-1] if names else "")


@pytest.mark.parametrize("k", [0, 1, 2])
def test_product_group_decoder_against_the_oracle_on_arbitrary_groups(pkg, oracle, k):
    """20 000 arbitrary groups per style (tools/ref_crosscheck.py::fuzz_group_lists: noise; one station with counting
    segment addresses and text bytes incl. the control characters the text decoders look for; the same with the
    station changing) through the product's host group decoder and the oracle's: the same frames, the same name.
    (`tools/ref_crosscheck.py --fuzz-groups 12 --only-groups` ran twelve such lists of 50 000 through the
    REFERENCE'S DecodeRDS in the build container: oracle == reference on all 600 000 groups / 994 000 frames.)"""
    from tools.ref_crosscheck import fuzz_group_lists
    name, groups = fuzz_group_lists(k + 1, n=20000)[k]
    L = oracle.lib()
    L.fmo_debug_push_group.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    o = oracle.OracleDecoder(2.4e6, -0.36e6, 48000.0, 15000.0, 11)
    gd = pkg.GroupDecoder()
    for g in groups:
        row = [int(x) for x in g]
        L.fmo_debug_push_group(o._h, (ctypes.c_uint16 * 4)(*row))
        gd.push(row)
    fr = o.uecp_frames()
    assert len(fr) > 20000 and gd.frames == fr, name
    assert gd.name[:8] == o.channel_name()[:8]


def _fir_blocks(fmsig, d):
    p = fmsig.default_params(d["fs"], noise_sigma=0.01, seed=d["seed"])
    blocks, pos, sha = [], 0, hashlib.sha256()
    for n in d["calls"]:
        b = fmsig.generate_f32(p, pos, n)
        sha.update(np.ascontiguousarray(b, dtype=np.float32).tobytes())
        blocks.append(b)
        pos += n
    assert sha.hexdigest() == d["iq_sha256"], "the signal generator no longer produces the fixture's input"
    return blocks


def _sha(x):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(x).view(np.uint32).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("case", FIR_CASES, ids=[c[0]["name"].replace(" ", "_")[:40] for c in FIR_CASES])
def test_oracle_if_stage_reproduces_the_reference_records(oracle, fmsig, case):
    d, sha, cnt = case
    o = oracle.OracleDecoder(d["fs"], -0.15 * d["fs"], 48000.0, 15000.0, d["D"], table_size=d["table"],
                             if_filter_order=0 if d["order"] == 8 * d["D"] else d["order"], tuning_shift=d["shift"])
    for k, b in enumerate(_fir_blocks(fmsig, d)):
        o.process_stream(b)
        y = o.taps()["demod"]
        assert y.size == int(cnt[k]), (k, y.size, int(cnt[k]))
        assert np.array_equal(_sha(y), sha[k]), "call %d: FIR output differs from the reference's" % k


def _by_geometry():
    """Cases that differ only in the tuning shift share their input: one batch, one channel per shift, one
    shared capture -- BASELINE configs[2]'s shape."""
    groups = {}
    for d, sha, cnt in FIR_CASES:
        key = (d["fs"], d["D"], d["table"], d["order"], d["seed"], tuple(d["calls"]))
        groups.setdefault(key, []).append((d, sha, cnt))
    return list(groups.values())


@pytest.mark.gpu
@pytest.mark.parametrize("group", _by_geometry(), ids=lambda g: g[0][0]["name"].split(",")[0].replace(" ", "_")[:32])
def test_hip_if_stage_reproduces_the_reference_records(fmsig, group):
    pkg = load_package()
    d0 = group[0][0]
    shifts = [d["shift"] for d, _, _ in group]
    params = pkg.make_params(d0["fs"], -0.15 * d0["fs"], 48000.0, 15000.0, d0["D"], table_size=d0["table"],
                             if_filter_order=0 if d0["order"] == 8 * d0["D"] else d0["order"])
    b = pkg.Batch(params, len(group), tuning_shifts=shifts, record_callbacks=False)
    b.enable_taps(True)
    assert min(d0["calls"]) >= b.min_samples()
    for k, blk in enumerate(_fir_blocks(fmsig, d0)):
        b.process_host(blk.view(np.complex64), shared=True)
        for c, (d, sha, cnt) in enumerate(group):
            y = b.tap("demod", c)
            assert y.size == int(cnt[k]), (k, c, y.size, int(cnt[k]))
            assert np.array_equal(_sha(y), sha[k]), "call %d, shift %d: FIR output differs from the reference's" % (
                k, d["shift"])
    b.close()
