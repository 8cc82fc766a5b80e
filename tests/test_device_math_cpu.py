"""csrc/fmd_math.h compiled for the host and swept against the functions the reference's CPU
build uses: glibc atan2f (the fdlibm restatement must be bit-identical) and the x87 fsincos
instruction (the FP64-evaluate-round-once sin/cos may differ only in double-rounding cases,
probability ~2^-28 per value).  The GPU executes the same source."""
import os
import re
import subprocess

from __graft_entry__ import ROOT


def test_math_restatements_match_host_libm(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "fmd_math_check")
    src = exe + ".c"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", src, "-lm", "-o", exe])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True, check=True).stdout
    m = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", out)}
    assert m["n"] == 20000000
    assert m["atan2f"] == 0, out        # literal fdlibm restatement
    assert m["atan2f_tab"] == 0, out    # table-parameterised form used in the kernels
    assert m["sincos_nco"] <= 2, out    # 40M values: expected ~0.15 double-rounding cases
    assert m["sincos_tab"] <= 2, out
    assert m["sincos_p256"] <= 2, out   # exact float reduction (the serial stage's two NCOs)
    assert m["u8_to_f32"] == 0, out     # RTL-SDR byte conversion, all 256 inputs
