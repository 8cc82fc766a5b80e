"""CPU-only checks of the product library: it loads, exports every symbol include/fmd.h declares,
fails loudly when no GPU is present (no CPU fallback), and its host-only pieces (UECP group
decoder, byte stuffing) match the oracle and the reference's recorded frames."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

REF_UECP = [  # frames the reference handed to AddUECPDataFrame
    "00 00 00 05 01 00 01 14 D3 0D 44",
    "00 00 01 04 07 00 01 0A C1 31",
    "00 00 02 04 03 00 01 00 64 6A",
    "00 00 03 04 05 00 01 01 16 72",
    "00 00 04 04 04 00 01 00 B8 A6",
    "00 00 05 0B 02 00 01 54 45 53 54 46 4D 30 31 D3 49",
]


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def test_every_declared_symbol_is_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "fmd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fmd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = pkg.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(pkg.EXPORTS) <= declared


def test_gather_library_exports_what_its_header_declares(pkg):
    """libfmd_gather.so (the whole-node step: rank-0 gather over RCCL, C++ above the two C ABIs) loads
    beside torch's RCCL and exports every symbol include/fmd_gather.h declares; no call is made (no
    communicator without a GPU)."""
    import importlib
    g = importlib.import_module(pkg.__name__ + ".gather")
    hdr = open(os.path.join(ROOT, "include", "fmd_gather.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fmd_gather_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 8
    lib = g.lib()
    assert not [n for n in sorted(declared) if not hasattr(lib, n)]
    assert set(g.EXPORTS) == declared
    assert os.path.exists(os.path.join(ROOT, "tools", "node_bench"))  # built by __graft_entry__.build()


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.FmdError, match="no HIP device|CPU fallback|hip"):
        pkg.Batch(pkg.make_params(2.4e6, -0.36e6, 48000.0, 15000.0, 11), 1)
    with pytest.raises(pkg.FmdError):
        pkg.FmDecoder(2.4e6, -0.36e6, 48000.0, 15000.0, 11)


def test_product_does_not_touch_the_oracle():
    """The product path must not import, link or load anything under oracle/."""
    pkgdir = os.path.join(ROOT, "pvr.rtl.radiofm_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                code = re.sub(r'""".*?"""', "", text, flags=re.S)
                code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)
                code = "\n".join(l for l in code.splitlines()
                                 if not l.strip().startswith(("#", "//", "*")))
                assert "fmd_oracle" not in code and "oracle_py" not in code, (f, "references oracle")
                assert "libfmd_oracle" not in code
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkgdir, "libfmd_hip.so")], capture_output=True,
                         text=True).stdout
    assert "oracle" not in out


def test_group_decoder_matches_reference_frames_and_oracle(pkg, oracle, fmsig):
    # groups as the oracle's signal path produces them for the SURVEY test signal
    fs = 2.4e6
    p = fmsig.default_params(fs)
    o = oracle.OracleDecoder(fs, -0.15 * fs, 48000.0, 15000.0, 11)
    for b in range(60):
        o.process_stream(fmsig.generate_f32(p, b * 65536, 65536))
    groups = [g for _, g in o.rds_groups()]
    assert len(groups) >= 12
    gd = pkg.GroupDecoder()
    for g in groups:
        gd.push(g)
    assert gd.frames == o.uecp_frames()
    assert [f.hex(" ").upper() for f in gd.frames[:6]] == REF_UECP
    assert gd.name == "TESTFM01"


def _rds_block_b(group, ver_b, pty=5, low5=0):
    return (group << 12) | (int(ver_b) << 11) | (pty << 5) | low5


def test_group_decoder_all_types_match_oracle(pkg, oracle):
    """Synthetic groups of every decoded type (0A/0B, 1A, 2A/2B, 3A RT+/TFC, 4A, 8A, 10A, ODA
    carriers, EON) through the product's host decoder and through the oracle's restatement."""
    import ctypes
    L = oracle.lib()
    rng = np.random.default_rng(5)
    groups = []
    pi = 0xABCD
    text = b"Hello from the MI355X radiotext test, 64 characters long okay!!"
    for rep in range(2):
        for seg in range(4):
            groups.append((pi, _rds_block_b(0, rep, low5=(seg | 0x08 | (0x04 if seg == 3 else 0))),
                           0xE0CD, int.from_bytes(b"PSNAME%02d"[:8][2 * seg:2 * seg + 2] if False
                                                  else (b"STATION%d" % rep)[2 * seg:2 * seg + 2], "big")))
        groups.append((pi, _rds_block_b(1, 0), 0x80E0, 0x1234 + rep))
        groups.append((pi, _rds_block_b(3, 0, low5=0x16), 0x0000, 0x4BD7))  # 3A: RT+ on 11A
        groups.append((pi, _rds_block_b(3, 0, low5=0x18), 0x1234, 0xCD46))  # 3A: TFC on 12A
        for seg in range(16):
            ab = rep << 4
            groups.append((pi, _rds_block_b(2, 0, low5=ab | seg),
                           int.from_bytes(text[4 * seg:4 * seg + 2], "big"),
                           int.from_bytes(text[4 * seg + 2:4 * seg + 4], "big")))
        groups.append((pi, _rds_block_b(2, 0, low5=(rep << 4) | 0), 0x4865, 0x6C6C))
        groups.append((pi, _rds_block_b(11, 0, low5=3), 0x2222, 0x3333))   # RT+ ODA payload
        groups.append((pi, _rds_block_b(12, 0, low5=1), 0x4444, 0x5555))   # TFC ODA payload
        groups.append((pi, _rds_block_b(4, 0, low5=0x01), 0xCF51, 0x2C40))  # clock-time
        groups.append((pi, _rds_block_b(8, 0, low5=9), 0xAAAA, 0xBBBB))     # TMC
        groups.append((pi, _rds_block_b(10, 0, low5=rep), 0x4A41, 0x5A5A))  # PTYN
        groups.append((pi, _rds_block_b(14, 0), 0x1111, 0x2222))            # EON: ignored
        groups.append((pi, _rds_block_b(2, 1, low5=5), 0x0000, 0x4142))     # 2B
        groups.append((pi, _rds_block_b(5, 0), 1, 2))                       # TDC: ignored
    for _ in range(40):  # random groups, same PI so state carries over
        groups.append((pi, int(rng.integers(0, 65536)), int(rng.integers(0, 65536)),
                       int(rng.integers(0, 65536))))
    groups.append((0x1234, _rds_block_b(0, 0), 0, 0x4142))  # PI change resets the decoder

    gd = pkg.GroupDecoder()
    for g in groups:
        gd.push(g)

    # the oracle's group decoder is reached through a decoder object; feed it directly
    L.fmo_debug_push_group.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    o = oracle.OracleDecoder(2.4e6, -0.36e6, 48000.0, 15000.0, 11)
    for g in groups:
        arr = (ctypes.c_uint16 * 4)(*g)
        L.fmo_debug_push_group(o._h, arr)
    ref = o.uecp_frames()
    assert len(ref) > 30
    assert gd.frames == ref


def test_uecp_byte_stuffing(pkg):
    # RadioReceiver.cpp:387-414: 0xFE start, 0xFF stop, 0xFD/0xFE/0xFF -> 0xFD (v&3)-1
    frame = bytes([0x00, 0xFD, 0x10, 0xFE, 0xFF, 0x7F])
    assert pkg.stuff_uecp_frame(frame) == bytes([0xFE, 0x00, 0xFD, 0x00, 0x10, 0xFD, 0x01, 0xFD, 0x02,
                                                 0x7F, 0xFF])


def test_cpp_header_drops_into_reference_declaration_order():
    """include/fm_decoder.hpp against the reference's own declaration order: `class cFmDecoder;`
    (RadioReceiver.h:23) and a `cFmDecoder*` member (:124) come BEFORE the decoder header, then
    `new cFmDecoder(this, ...)` (RadioReceiver.cpp:296-300), Reset, ProcessStream, the getters and
    `delete` (:373).  Compile-only (g++ -fsyntax-only), C++14 like the reference's CMakeLists."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "dropin_decl_order.cpp")
    for std in ("-std=c++14", "-std=c++17"):
        out = subprocess.run(["g++", std, "-fsyntax-only", "-Wall", "-Werror",
                              "-I" + os.path.join(ROOT, "include"), src],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
    # and the same header where cRadioReceiver is only defined AFTER the include (receiver_demo.cpp)
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "cpp", "receiver_demo.cpp")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_the_library_reads_no_environment_variable():
    """Development switches are per batch and by API (fmd_batch_debug_set): the shipped library neither
    imports getenv nor carries the name of a switch."""
    import subprocess
    from __graft_entry__ import PKG_DIR
    so = os.path.join(PKG_DIR, "libfmd_hip.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    names = subprocess.run(["strings", so], capture_output=True, text=True, check=True).stdout
    # (error messages may name a constant of include/fmd.h; nothing else that looks like a variable)
    declared = set(re.findall(r"FMD_[A-Z0-9_]+", open(os.path.join(ROOT, "include", "fmd.h")).read()))
    assert set(re.findall(r"FMD_[A-Z0-9_]+", names)) <= declared


def test_development_keys_are_the_documented_ones():
    """fmd_batch_debug_set takes the keys INTEGRATION.md section 3 lists, each with the test that exercises it -- no
    more (round 6 cut 32 keys to 14), and every test the table names exists."""
    src = open(os.path.join(ROOT, "pvr.rtl.radiofm_amd", "csrc", "fmd_batch.hip")).read()
    body = src[src.index("int fmd_batch_debug_set(fmd_batch* b, const char* key, int value)"):]
    body = body[:body.index("\nint fmd_batch_wait(")]
    keys = set(re.findall(r'k == "([a-z_0-9]+)"', body))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("| key | values | what it selects | exercised by |"):]
    table = table[:table.index("\n\n")]
    documented = set()
    for row in table.splitlines()[2:]:
        cells = [c.strip() for c in row.strip("|").split("|")]
        documented |= set(re.findall(r"`([a-z_0-9]+)`", cells[0]))
        for path, name in re.findall(r"`(tests/[a-z_0-9]+\.py|test_[a-z_0-9]+\.py)(?:::([a-z_0-9]+))?`", cells[-1]):
            path = path if path.startswith("tests/") else "tests/" + path
            text = open(os.path.join(ROOT, path)).read()
            assert not name or ("def %s(" % name) in text, (path, name)
    assert keys == documented, (sorted(keys - documented), sorted(documented - keys))
    assert len(keys) <= 14
