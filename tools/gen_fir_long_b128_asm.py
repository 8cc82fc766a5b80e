#!/usr/bin/env python3
"""Generates the hand-scheduled long-filter tap loop of k_if_fir for the PLAIN window (E = 0) and an
even decimation D = 2 * odd, reading TWO consecutive samples per lane with one ds_read_b128, as the
inline-asm body of fir_long_b128_asm in pvr.rtl.radiofm_amd/csrc/fmd_k_if.hip.h.

    python tools/gen_fir_long_b128_asm.py     (paste the output between `asm volatile(` and `);`)

With the window of a long filter only ONE wave fits a SIMD, and a lone wave issues a packed f32
instruction every 8 cycles (tools/ubench/pk_rate: v_pk_mul_f32 / v_pk_add_f32 8.0 cycles per
instruction and wave at one wave per SIMD, whatever depends on what): the two packed instructions a
tap needs without FMA cost 16 cycles, and every other instruction of the loop comes on top.  So the
loop is written for the FEWEST instructions per tap: ds_read_b128 fetches the samples of taps
(j + 1, j) -- one LDS instruction per two taps, at 256 B/clk/CU where ds_read2_b64 delivers 128
(MI355X_MICROARCH.md, LDS table) -- and one s_load_dwordx16 the 16 taps of a batch.  The lanes of a
wave read samples D apart; with D = 2 * odd the 16 lanes of a b128 lane group start on 16 different
4-bank groups: conflict-free without de-interleaving the window.

Two register sets (A, B) of 16 taps = 8 reads: v[base + 4i : base + 4i + 3] = samples of taps
(j + 2i + 1, j + 2i) -- the lower address is the OLDER sample, i.e. the higher tap.  Per half
iteration: wait, issue the other set's loads, then this set's arithmetic with the products two ahead
of the running sum; the sum is ONE chain in tap order.
(Measured, 4096 taps, D = 46, 4096 channels, same box: 3.35 ms per launch; the two-region loop with
ds_read2_b64, fir_long_e1_asm, 3.51 ms; a variant with the taps staged in LDS too, so that every
wait is a counted one and reads run four groups ahead, 3.54 ms -- more instructions per tap.)"""
A, B = 64, 96        # sample sets: v[64:95], v[96:127]
TMP = 128            # v[128:135]: four product pairs
KA, KB = 40, 56      # taps: s[40:55], s[56:71]
lines = []
def emit(s): lines.append(s)
def load(v, k):
    emit(f"s_load_dwordx16 s[{k}:{k+15}], s[72:73], 0x0")
    for i in range(8):  # read i holds taps (2i+1, 2i) of the batch: 16 bytes, i pairs below the first
        emit(f"ds_read_b128 v[{v+4*i}:{v+4*i+3}], %1 offset:{16*(7-i)}")
    emit("v_subrev_u32 %1, 128, %1")
    emit("s_add_u32 s72, s72, 64")
    emit("s_addc_u32 s73, s73, 0")
def mul(t, v, k):
    i = t // 2
    tp = TMP + 2*(t % 4)
    kp = k + 2*i
    if t % 2 == 0:   # tap 2i: the newer sample = upper half of the read, tap in the low SGPR of the pair
        emit(f"v_pk_mul_f32 v[{tp}:{tp+1}], v[{v+4*i+2}:{v+4*i+3}], s[{kp}:{kp+1}] op_sel_hi:[1,0]")
    else:            # tap 2i+1: the older sample = lower half, tap in the high SGPR
        emit(f"v_pk_mul_f32 v[{tp}:{tp+1}], v[{v+4*i}:{v+4*i+1}], s[{kp}:{kp+1}] op_sel:[0,1]")
def add(t):
    tp = TMP + 2*(t % 4)
    emit(f"v_pk_add_f32 %0, %0, v[{tp}:{tp+1}]")
def mac(v, k):
    mul(0, v, k); mul(1, v, k)
    for t in range(16):
        add(t)
        if t + 2 < 16:
            mul(t + 2, v, k)
        elif t == 14:
            emit("s_nop 0")
emit("s_mov_b32 s72, %3")
emit("s_mov_b32 s73, %4")
load(A, KA)
emit("1:")
emit("s_waitcnt lgkmcnt(0)")
load(B, KB)
mac(A, KA)
emit("s_waitcnt lgkmcnt(0)")
load(A, KA)
mac(B, KB)
emit("s_sub_u32 %2, %2, 1")
emit("s_cmp_lg_u32 %2, 0")
emit("s_cbranch_scc1 1b")
emit("s_waitcnt lgkmcnt(0)")
print("\n".join('      "%s\\n\\t"' % l for l in lines))
print('      : "+v"(acc2), "+v"(a), "+s"(cnt)')
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
