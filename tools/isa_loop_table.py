#!/usr/bin/env python3
"""Instruction count of the serial stage's two sample loops (k_demod_serial), by class, from the ISA
hipcc emits for gfx950.  The stage is bound by what ONE wave per SIMD can issue, so
the instruction count of these loops is its cost model.

    python tools/isa_loop_table.py [extra hipcc -D flags ...]      -> table on stdout

The FM wave's loop is the innermost loop with the IEEE division expansion (v_div_scale_f32) that
writes one LDS word per sample; the second wave's is the one with the two global stores."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "serial_stage.hip")


def classify(op):
    if op.startswith(("s_nop", "s_waitcnt")):
        return "wait / nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "scalar ALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global memory"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "VALU transcendental (quarter rate)"
    if op.startswith("v_cvt") and "f64" in op:
        return "VALU FP64 convert"
    if "f64" in op or op.startswith("v_ldexp_f64"):
        return "VALU FP64"
    if op.startswith("v_pk_"):
        return "VALU FP32 packed"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "VALU FP32 division helpers"
    if op.startswith("v_cmp"):
        return "VALU compare"
    if op.startswith(("v_cndmask", "v_mov", "v_readfirstlane", "v_perm", "v_bfe", "v_bfi", "v_and", "v_or", "v_xor",
                      "v_lshl", "v_lshr", "v_ashr", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add3", "v_add_co",
                      "v_addc", "v_mad_u", "v_mul_u", "v_mul_lo", "v_med3_u", "v_med3_i", "v_min_u", "v_max_u",
                      "v_bitop", "v_alignbit", "v_cvt_f32_ubyte", "v_cvt_f32_u", "v_cvt_f32_i", "v_cvt_u32", "v_cvt_i32",
                      "v_lshl_add", "v_lshl_or", "v_and_or", "v_accvgpr")):
        return "VALU integer / move / select"
    if op.startswith("v_"):
        return "VALU FP32"
    return "other"


def loops_of(fn_lines):
    """[(header label, [instruction lines])] for every loop closed by a backward branch"""
    labels = {}
    for i, l in enumerate(fn_lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(fn_lines):
        m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            body = [x for x in fn_lines[labels[m.group(2)]:i + 1] if re.match(r"^\s+[a-z]", x)]
            out.append((m.group(2), body))
    return out


def main():
    flags = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=off", "-S", "--cuda-device-only", SRC, "-o", asm] + flags,
                              stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    start = [i for i, l in enumerate(text) if re.match(r"^_ZN3fmd14k_demod_serialILi2ELb1E", l)][0]
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
    loops = loops_of(text[start:end])
    # candidates: innermost loops with the division expansion; samples per trip = LDS words written
    # (FM wave) / pairs of global stores (second wave); the loop with the most samples per trip is the
    # one full chunks run (the other is the rolled loop of a ragged last chunk)
    fm = second = None
    fm_n = second_n = 0
    spans = {lab: (len(body)) for lab, body in loops}
    for lab, body in loops:
        ops = [b.split()[0] for b in body]
        # innermost loops only: no other loop's body is contained in this one
        if any(l2 != lab and len(b2) < len(body) and b2[0] in body and b2[-1] in body for l2, b2 in loops):
            continue
        ndiv = sum(o.startswith("v_div_fixup_f32") for o in ops)
        if not ndiv:
            continue
        nst = sum(o.startswith("global_store") for o in ops)
        nw = sum(o.startswith("ds_write_b32") for o in ops)
        if nst and nst // 2 == ndiv and (nst // 2 > second_n):
            second, second_n = body, nst // 2
        elif not nst and nw == ndiv and nw > fm_n:
            fm, fm_n = body, nw
    # The FM wave's full-chunk loop runs its group of samples as ONE basic block (a single rare-input
    # test per group, behind it): the longest run of instructions without a label or a branch that
    # writes LDS words and divides but does not store to global memory.
    blocks, cur = [], []
    for l in text[start:end]:
        if re.match(r"^\.LBB", l) or re.match(r"^\s+(s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm)", l):
            if re.match(r"^\s+s_", l):
                cur.append(l)
            blocks.append(cur)
            cur = []
        elif re.match(r"^\s+[a-z]", l):
            cur.append(l)
    best = None
    for blk in blocks:
        ops = [b.split()[0] for b in blk]
        nd = sum(o.startswith("v_div_fixup_f32") for o in ops)  # the first division of every sample's arctangent
        if nd >= 2 and not any(o.startswith("global_store") for o in ops):
            if best is None or nd > best[1]:
                best = (blk, nd)
    fm_note = ""
    if best and best[1] > fm_n:
        fm, fm_n = best
        fm_note = ("  (the group's straight-line block; its %d LDS writes and the loop latch, ~3 instructions per "
                   "sample, sit behind the group's rare-input test)" % fm_n)
    print("flags:", " ".join(flags) or "(none)")
    for name, body, n in (("FM wave (FM PLL: sincos, complex product, atan2f, loop update)", fm, fm_n),
                          ("second wave (DC filter, pilot PLL, 38 kHz product, RDS oscillator, stores)", second,
                           second_n)):
        if body is None:
            print(name, ": loop not found")
            continue
        cnt = collections.Counter(classify(b.split()[0]) for b in body)
        total = sum(cnt.values())
        print("\n%s: %d instructions per trip of %d sample%s = %.1f per sample (common path)%s"
              % (name, total, n, "" if n == 1 else "s", total / n, fm_note if body is fm else ""))
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
            print("  %-40s %6.1f" % (k, v / n))
        valu = sum(v for k, v in cnt.items() if k.startswith("VALU"))
        print("  %-40s %6.1f" % ("= vector ALU", valu / n))


if __name__ == "__main__":
    main()
