#!/usr/bin/env python3
"""Generates the hand-scheduled long-filter tap loops of k_if_fir for the window layouts that are read
one sample (8 bytes) at a time, as the inline-asm bodies of

    fir_long_odd_asm   G = 1   plain window, odd decimation D (a 16-byte read would be misaligned for
                               every other lane)
    (G = 2 / 4 without `b128`: the two- / four-region windows read 8 bytes at a time -- fir_long_e1_asm /
     fir_long_e2_asm, shipped until round 5 behind a development key, no longer part of the library)

in pvr.rtl.radiofm_amd/csrc/fmd_k_if.hip.h (D = 2 * odd normally runs fir_long_b128_asm,
tools/gen_fir_long_b128_asm.py).

    python tools/gen_fir_long_asm.py G [b128]   (paste the output between `asm volatile(` and `);`)

With `b128` (G = 2, 4 only) the two adjacent positions a ds_read2_b64 would fetch come with ONE
ds_read_b128 -- twice the LDS rate (256 instead of 128 B/clk/CU).  That needs every lane's pair on a
16-byte boundary: the lane stride inside a region, D / G, must be even, i.e. it is the form for
D = 4 * odd in the TWO-region window (fir_long_e1_b128_asm) and D = 8 * odd in the four-region window
(fir_long_e2_b128_asm), with the region size even and the batch's lowest position even (the kernel puts
one round of taps in front when it is not).

A batch is 16 taps.  In a G-region window tap j + s (s = 0 .. G-1) of a round sits in region
G - 1 - s, all at the same position, one position lower per round; a batch therefore takes 16 / G
consecutive positions from each region: two per ds_read2_b64 (eight LDS instructions per batch in every
layout), and the 16 taps come with one s_load_dwordx16.  Two register sets (A, B): per half iteration
wait, issue the other set's loads, then this set's arithmetic with the products two ahead of the running
sum; the sum is ONE chain in tap order.  Operands: %0 the running (re, im) sum; %1 .. %G the LDS byte
address of the LOWEST position of the current batch in the region of tap j + s; then the count of
batch pairs (32 taps each, >= 1) and the address of the batch's first tap (lo, hi).  The last batch load
is a dummy: 16 taps past the table (padded) and 16 / G positions below the last batch (region 0's lie in
the 32 slots the kernel keeps in front of the window; the other regions' in the region below).
(A variant with the tap table in LDS was slower: the loop is LDS-bandwidth-bound.)"""
import sys
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B128 = len(sys.argv) > 2 and sys.argv[2] == "b128"
assert G in (1, 2, 4) and not (B128 and G == 1)
NPOS = 16 // G       # positions a batch takes from each region
A, B = 64, 96        # sample sets: v[64:95], v[96:127]; the region of tap j + s at base + s * 2 * NPOS
TMP = 128            # v[128:135]: four product pairs
KA, KB = 40, 56      # taps: s[40:55], s[56:71]
CNT, KLO, KHI = G + 1, G + 2, G + 3
lines = []
def emit(s): lines.append(s)
def load(v, k):
    emit(f"s_load_dwordx16 s[{k}:{k+15}], s[72:73], 0x0")
    for s in range(G):
        base = v + s * 2 * NPOS
        for q in range(NPOS // 2):  # positions r = 2q, 2q + 1 below the round's first: the higher address first
            o0, o1 = NPOS - 1 - 2*q, NPOS - 2 - 2*q
            off = f"offset0:{o0} offset1:{o1}" if o1 else f"offset0:{o0}"
            if B128:  # the pair's lower position first: v[+0:+1] = r = 2q + 1, v[+2:+3] = r = 2q
                emit(f"ds_read_b128 v[{base+4*q}:{base+4*q+3}], %{1+s} offset:{8*o1}")
            else:
                emit(f"ds_read2_b64 v[{base+4*q}:{base+4*q+3}], %{1+s} {off}")
    for s in range(G):
        emit(f"v_subrev_u32 %{1+s}, {8*NPOS}, %{1+s}")
    emit("s_add_u32 s72, s72, 64")
    emit("s_addc_u32 s73, s73, 0")
def mul(t, v, k):
    s, r = t % G, t // G
    x = v + s * 2 * NPOS + 2 * r
    if B128:
        x = v + s * 2 * NPOS + 4 * (r // 2) + (0 if r % 2 else 2)
    tp = TMP + 2*(t % 4)
    kp = k + 2*(t // 2)
    sel = "op_sel_hi:[1,0]" if t % 2 == 0 else "op_sel:[0,1]"
    emit(f"v_pk_mul_f32 v[{tp}:{tp+1}], v[{x}:{x+1}], s[{kp}:{kp+1}] {sel}")
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
emit(f"s_mov_b32 s72, %{KLO}")
emit(f"s_mov_b32 s73, %{KHI}")
load(A, KA)
emit("1:")
emit("s_waitcnt lgkmcnt(0)")
load(B, KB)
mac(A, KA)
emit("s_waitcnt lgkmcnt(0)")
load(A, KA)
mac(B, KB)
emit(f"s_sub_u32 %{CNT}, %{CNT}, 1")
emit(f"s_cmp_lg_u32 %{CNT}, 0")
emit("s_cbranch_scc1 1b")
emit("s_waitcnt lgkmcnt(0)")
print("\n".join('      "%s\\n\\t"' % l for l in lines))
addrs = ", ".join('"+v"(a%d)' % (G - 1 - s) for s in range(G))
print(f'      : "+v"(acc2), {addrs}, "+s"(cnt)')
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
