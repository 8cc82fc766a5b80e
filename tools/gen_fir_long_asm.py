#!/usr/bin/env python3
"""Generates the hand-scheduled long-filter tap loop of k_if_fir (E = 1 window layout) as the
inline-asm body of fir_long_e1_asm in pvr.rtl.radiofm_amd/csrc/fmd_kernels.hip.h.

    python tools/gen_fir_long_asm.py      (paste the output between `asm volatile(` and `);`)

Two register sets (A, B) of 16 taps: 8 samples from region 1, 8 from region 0 (ds_read2_b64) and
the 16 taps by one s_load_dwordx16.  Per half iteration: wait, issue the other set's loads, then
this set's arithmetic with the products two ahead of the running sum.  (A variant with the tap
table in LDS was slower: the loop is LDS-bandwidth-bound.)"""
A1, A0 = 64, 80      # set A: region-1 samples v[64:79], region-0 samples v[80:95]
B1, B0 = 96, 112     # set B
TMP = 128            # v[128:135]: four product pairs
KA, KB = 40, 56      # taps: s[40:55], s[56:71]
lines = []
def emit(s): lines.append(s)
def load(v1, v0, k):
    emit(f"s_load_dwordx16 s[{k}:{k+15}], s[72:73], 0x0")
    for q in range(4):
        o0, o1 = 7 - 2*q, 6 - 2*q
        off = f"offset0:{o0} offset1:{o1}" if o1 else f"offset0:{o0}"
        emit(f"ds_read2_b64 v[{v1+4*q}:{v1+4*q+3}], %1 {off}")
    for q in range(4):
        o0, o1 = 7 - 2*q, 6 - 2*q
        off = f"offset0:{o0} offset1:{o1}" if o1 else f"offset0:{o0}"
        emit(f"ds_read2_b64 v[{v0+4*q}:{v0+4*q+3}], %2 {off}")
    emit("v_subrev_u32 %1, 64, %1")
    emit("v_subrev_u32 %2, 64, %2")
    emit("s_add_u32 s72, s72, 64")
    emit("s_addc_u32 s73, s73, 0")
def mul(t, v1, v0, k):
    r = t // 2
    tp = TMP + 2*(t % 4)
    kp = k + 2*r
    if t % 2 == 0:
        emit(f"v_pk_mul_f32 v[{tp}:{tp+1}], v[{v1+2*r}:{v1+2*r+1}], s[{kp}:{kp+1}] op_sel_hi:[1,0]")
    else:
        emit(f"v_pk_mul_f32 v[{tp}:{tp+1}], v[{v0+2*r}:{v0+2*r+1}], s[{kp}:{kp+1}] op_sel:[0,1]")
def add(t):
    tp = TMP + 2*(t % 4)
    emit(f"v_pk_add_f32 %0, %0, v[{tp}:{tp+1}]")
def mac(v1, v0, k):
    mul(0, v1, v0, k); mul(1, v1, v0, k)
    for t in range(16):
        add(t)
        if t + 2 < 16:
            mul(t + 2, v1, v0, k)
        elif t == 14:
            emit("s_nop 0")
emit("s_mov_b32 s72, %4")
emit("s_mov_b32 s73, %5")
load(A1, A0, KA)
emit("1:")
emit("s_waitcnt lgkmcnt(0)")
load(B1, B0, KB)
mac(A1, A0, KA)
emit("s_waitcnt lgkmcnt(0)")
load(A1, A0, KA)
mac(B1, B0, KB)
emit("s_sub_u32 %3, %3, 1")
emit("s_cmp_lg_u32 %3, 0")
emit("s_cbranch_scc1 1b")
emit("s_waitcnt lgkmcnt(0)")
print("\n".join('      "%s\\n\\t"' % l for l in lines))
print('      : "+v"(acc2), "+v"(a1), "+v"(a0), "+s"(cnt)')
print('      : "s"(klo), "s"(khi)')
clob = ['"v%d"' % i for i in range(64, 136)] + ['"s%d"' % i for i in range(40, 74)] + ['"scc"', '"memory"']
out, line = [], "      : "
for c in clob:
    if len(line) + len(c) + 2 > 100:
        out.append(line.rstrip())
        line = "        "
    line += c + ", "
out.append(line.rstrip().rstrip(","))
print("\n".join(out))
