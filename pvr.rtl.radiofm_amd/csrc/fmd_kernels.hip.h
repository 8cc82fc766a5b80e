/*
 * fmd_kernels.hip.h -- gfx950 kernels of the batched FM decoder.
 *
 * Data layout in HBM (C channels, CP = C rounded up to 64):
 *   IQ input        [C][N]      complex<float>, one contiguous stream per channel (API layout)
 *   demod           [C][Mstride] complex<float>  channel-major (written coalesced by the FIR)
 *   everything else [row][CP]   "time-major": one row per sample instant, channels contiguous,
 *                               so a wavefront = 64 channels at one instant and every access is a
 *                               coalesced 256/512-byte row segment.  Buffers that feed a windowed
 *                               stage start with H "history" rows (the last H samples of the
 *                               previous call), so a window never needs a branch.
 * All positions (decimator phase, resampler fraction, tuner index, FIR ring index) are the same
 * for every channel of a batch and are tracked on the host; only signal state is per channel.
 *
 * Arithmetic is float with the reference's promotions, sequential accumulation in the
 * reference's order, and no FMA contraction (the file is compiled with -ffp-contract=off), so
 * the outputs are bit-comparable with the CPU path.  Citations: /root/reference/src/.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmd_math.h"

namespace fmd
{

struct DemodConsts
{
  // FM PLL (FmDecode.cpp:305-312, :254)
  float pll_alpha, pll_beta, nco_hl, nco_ll, demod_gain;
  // pilot PLL (FmDecode.cpp:88-140)
  float p_minfreq, p_maxfreq, p_b0, p_a1, p_a2, p_lf_b0, p_lf_b1, p_minsignal;
  int p_lock_delay;
  // RDS quadrature oscillator (DownConvert.cpp:311-320)
  float osc_cos, osc_sin;
};

struct RdsConsts
{
  float pll_alpha, pll_beta, nco_hl, nco_ll;
  float bs_b0, bs_b1, bs_b2, bs_a1, bs_a2; // bit-sync resonator
  int mf_taps;
};

struct AudioConsts
{
  float de_alpha;
  float n_b0, n_b1, n_b2, n_a1, n_a2; // 19 kHz notch
};

/* per-channel signal state: structure of arrays, every array CP long, addressed by slot
 * index from two slabs (one pointer each keeps the kernels' SGPR budget small) */
enum FSlot
{
  F_NCO_PHASE, F_NCO_INCR, F_DC_OFF,                 // FM PLL
  F_IF_LEVEL, F_BB_MEAN, F_BB_LEVEL,                 // level meters
  F_P_I1, F_P_I2, F_P_Q1, F_P_Q2, F_P_X1, F_P_FREQ, F_P_PHASE, F_P_LEVEL, // pilot PLL
  F_OSC_RE, F_OSC_IM,                                // RDS oscillator
  F_R_PHASE, F_R_FREQ, F_R_W1, F_R_W2, F_R_LAST_SYNC, F_R_LAST_SLOPE, F_R_LAST_DATA,
  F_DE_RE, F_DE_IM, F_N_W1A, F_N_W2A, F_N_W1B, F_N_W2B, // de-emphasis, notch
  F_AUDIO_MEAN, F_AUDIO_RMS, F_AUDIO_LEVEL,          // cRadioReceiver's audio level meter
  F_SLOTS
};
enum ISlot
{
  I_P_LOCK_CNT, I_STEREO,
  // the flag per call index mod 4, read by that call's audio tail: the serial stage of call k+4 is the
  // next writer of call k's copy, and it runs behind FIR(k+4), which waits for heavy(k+2), which waits
  // for the audio tail of call k (EV_AUD) -- the reuse is ordered by events, not by timing
  I_STEREO_Q0, I_STEREO_Q1, I_STEREO_Q2, I_STEREO_Q3,
  I_R_LAST_BIT, I_R_BITS, I_R_BLOCK, I_R_BITPOS, I_R_STATE, I_R_BOFF,
  I_R_ERRORS, I_R_SEQ, I_SLOTS
};
/* Device-side error words of a batch (host-mapped memory: the host reads them without a copy).
 * Kernels OR a bit in when an invariant fails.  err[0] holds the fatal conditions (the batch refuses
 * further calls until it is reset), err[1] the recoverable ones (reported once, then cleared). */
enum DevErr : unsigned
{
  DEVERR_SERIAL_HANDSHAKE = 1u, // err[0], k_demod_serial: a role wave gave up waiting for its partner
  DEVERR_RDS_QUEUE_FULL = 1u    // err[1], k_rds_bits / k_rds_export: a group did not fit (lost)
};
/* Status snapshot of every channel in host-mapped memory, [HS_WORDS][CP] 32-bit words: what the
 * cFmDecoder getters (FmDecode.h:140-165) and cRadioReceiver's audio meter return.  The kernels of a
 * call leave the record in device memory (ChannelState::ds: k_audio_tail, k_rds_bits); the last
 * kernel of the call (k_status_publish) copies it out so that the host reads it without touching the
 * device.  HS_SEQ_BEGIN is written first and HS_SEQ_END last (both = the call's index); a reader
 * takes END, the fields, then BEGIN, and has a consistent record when the two are equal. */
enum HostStatusWord
{
  HS_SEQ_BEGIN, HS_IF_LEVEL, HS_BB_MEAN, HS_BB_LEVEL, HS_P_LEVEL, HS_STEREO, HS_R_STATE,
  HS_AUDIO_MEAN, HS_AUDIO_RMS, HS_AUDIO_LEVEL, HS_SEQ_END, HS_WORDS
};
struct ChannelState
{
  float* f;         // [F_SLOTS][CP]
  int* i;           // [I_SLOTS][CP]
  uint16_t* r_data; // [4][CP]   block words of the group being assembled
  unsigned* err;    // the batch's two error words (DevErr)
  unsigned* hs;     // [HS_WORDS][CP] status snapshot in host-mapped memory (written by k_status_publish)
  unsigned* ds;     // [HS_WORDS][CP] the same record in device memory: what the kernels write
  unsigned spin_limit; // bound of the LDS hand-off waits (0 = every wait times out: test knob)
  unsigned CP;
  __host__ __device__ float* F(int slot) const { return f + (size_t)slot * CP; }
  __host__ __device__ int* I(int slot) const { return i + (size_t)slot * CP; }
};

struct RdsGroupRec
{
  uint32_t channel;
  uint32_t call_index;
  uint32_t seq;
  uint16_t blocks[4];
};

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits
 * for this wave's outstanding global STORES (1-2 us each time); the role-waves and tile loops
 * below only exchange data through LDS. */
/* s_setprio takes an immediate */
__device__ __forceinline__ void wave_prio(unsigned p)
{
  if (p == 1u)
    __builtin_amdgcn_s_setprio(1);
  else if (p == 2u)
    __builtin_amdgcn_s_setprio(2);
  else if (p == 3u)
    __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
/* Same for a single-wave workgroup: LDS operations of one wave execute in order, so only the
 * compiler has to be kept from reordering across the exchange. */
__device__ __forceinline__ void lds_wave_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

/* Two waves of one workgroup handing LDS buffers to each other without stopping the workgroup's other
 * waves at a barrier: a progress counter in LDS per direction.  LDS operations of a wave execute in
 * order, so the counter written after the data is seen after the data; the asm statements keep the
 * compiler from moving LDS accesses across.  The wait is bounded (~0.1 s): a protocol error does not hang
 * the device, it sets DEVERR_SERIAL_HANDSHAKE in the batch's error word. */
__device__ __forceinline__ void lds_publish(unsigned lds_addr, unsigned value)
{ // explicit DS instructions on the 32-bit LDS address: a generic pointer would make these FLAT
  // accesses, whose waits also drain the wave's global stores
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr), "v"(value) : "memory");
}
__device__ __forceinline__ void dev_error(unsigned* err, unsigned bit)
{ // system scope: the word lives in host-mapped memory
  __hip_atomic_fetch_or(err, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void lds_wait_ge(unsigned lds_addr, unsigned value, unsigned limit, unsigned* err)
{
  bool ok = false;
#pragma unroll 1
  for (unsigned spins = 0; spins < limit; spins++)
  {
    unsigned seen;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(lds_addr) : "memory");
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)seen) >= value)
    {
      ok = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  // gave up (~0.1 s with the default limit): the results of this call are wrong from here on; the
  // host learns it from the batch's error word (FMD_ERR_DEVICE from fmd_batch_wait / collect_rds)
  if (!ok && __builtin_amdgcn_readfirstlane((int)threadIdx.x) == (int)threadIdx.x)
    dev_error(err, DEVERR_SERIAL_HANDSHAKE);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  // std::complex<float> product: (ac - bd) + i(ad + bc), four products and two sums, each rounded
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

/* Hand-scheduled tap loop for long filters in the two-region window (E = 1), where one wave per
 * SIMD has to hide every LDS / scalar-cache round trip itself: batches of 16 taps in two register
 * sets, the 8 + 8 sample reads and the 16-tap scalar load of the next batch issued before the
 * arithmetic of the current one, one full wait per batch that only finds completed operations.
 * Products run two ahead of the sum; the sum itself stays one chain in tap order.  acc2 is the
 * running (re, im) sum; a1 / a0 are the LDS byte addresses of the lowest of the 8 positions of the
 * current batch in region 1 / region 0 (region 1 holds the first tap of every pair); klo / khi is
 * the address of the batch's first tap; cnt counts pairs of batches (32 taps each, >= 1).  The last
 * batch load is a dummy: it reads 16 taps past the table (the buffer is padded) and 8 positions
 * below the last batch (region 0's lie in the 32 slots the kernel keeps in front of the window).  Measured (4096 taps, D = 46):
 * 3.5 ms per launch against 5.8 ms for the compiler-scheduled loop; the LDS array delivers
 * ~110 B/clk/CU here, i.e. the loop is LDS-bandwidth-bound (a variant with the tap table in LDS
 * too: 4.1 ms).  Body generated by tools/gen_fir_long_asm.py. */
typedef float fmd_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fir_long_e1_asm(fmd_f2v& acc2, unsigned& a1, unsigned& a0, unsigned klo,
                                                unsigned khi, unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %4\n\t"
      "s_mov_b32 s73, %5\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[72:75], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[76:79], %1 offset0:1\n\t"
      "ds_read2_b64 v[80:83], %2 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[84:87], %2 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[88:91], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %2 offset0:1\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[96:99], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[100:103], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[104:107], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[108:111], %1 offset0:1\n\t"
      "ds_read2_b64 v[112:115], %2 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[116:119], %2 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[120:123], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[124:127], %2 offset0:1\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[64:65], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[80:81], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[66:67], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[82:83], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[68:69], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[84:85], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[70:71], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[86:87], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[72:73], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[88:89], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[74:75], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[90:91], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[76:77], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[92:93], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[78:79], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[94:95], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[72:75], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[76:79], %1 offset0:1\n\t"
      "ds_read2_b64 v[80:83], %2 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[84:87], %2 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[88:91], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %2 offset0:1\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[96:97], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[112:113], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[98:99], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[114:115], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[100:101], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[116:117], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[102:103], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[118:119], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[104:105], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[120:121], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[106:107], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[122:123], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[108:109], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[124:125], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[110:111], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[126:127], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %3, %3, 1\n\t"
      "s_cmp_lg_u32 %3, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a1), "+v"(a0), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory");
}

/* The same loop for the plain window and an ODD decimation (G = 1 of tools/gen_fir_long_asm.py): the 16
 * taps of a batch are 16 consecutive slots, read two at a time with ds_read2_b64 -- a 16-byte read
 * would be misaligned for every other lane (the lanes sit D slots apart).  a0 = LDS byte address of
 * tap j + 15 (the batch's lowest slot). */
__device__ __forceinline__ void fir_long_odd_asm(fmd_f2v& acc2, unsigned& a0, unsigned klo, unsigned khi,
                                                 unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %3\n\t"
      "s_mov_b32 s73, %4\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:15 offset1:14\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:13 offset1:12\n\t"
      "ds_read2_b64 v[72:75], %1 offset0:11 offset1:10\n\t"
      "ds_read2_b64 v[76:79], %1 offset0:9 offset1:8\n\t"
      "ds_read2_b64 v[80:83], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[84:87], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[88:91], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %1 offset0:1\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[96:99], %1 offset0:15 offset1:14\n\t"
      "ds_read2_b64 v[100:103], %1 offset0:13 offset1:12\n\t"
      "ds_read2_b64 v[104:107], %1 offset0:11 offset1:10\n\t"
      "ds_read2_b64 v[108:111], %1 offset0:9 offset1:8\n\t"
      "ds_read2_b64 v[112:115], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[116:119], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[120:123], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[124:127], %1 offset0:1\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[64:65], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[66:67], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[68:69], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[70:71], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[72:73], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[74:75], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[76:77], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[78:79], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[80:81], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[82:83], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[84:85], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[86:87], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[88:89], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[90:91], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[92:93], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[94:95], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:15 offset1:14\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:13 offset1:12\n\t"
      "ds_read2_b64 v[72:75], %1 offset0:11 offset1:10\n\t"
      "ds_read2_b64 v[76:79], %1 offset0:9 offset1:8\n\t"
      "ds_read2_b64 v[80:83], %1 offset0:7 offset1:6\n\t"
      "ds_read2_b64 v[84:87], %1 offset0:5 offset1:4\n\t"
      "ds_read2_b64 v[88:91], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %1 offset0:1\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[96:97], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[98:99], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[100:101], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[102:103], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[104:105], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[106:107], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[108:109], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[110:111], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[112:113], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[114:115], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[116:117], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[118:119], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[120:121], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[122:123], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[124:125], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[126:127], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %2, %2, 1\n\t"
      "s_cmp_lg_u32 %2, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a0), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory"
  );
}

/* And for the four-region window (D = 4 * odd and higher powers of two; G = 4): tap j + s sits in region
 * 3 - s, a batch takes four consecutive positions from each region.  a3 .. a0 = LDS byte address of the
 * lowest of them in region 3 .. 0. */
__device__ __forceinline__ void fir_long_e2_asm(fmd_f2v& acc2, unsigned& a3, unsigned& a2, unsigned& a1,
                                                unsigned& a0, unsigned klo, unsigned khi, unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %6\n\t"
      "s_mov_b32 s73, %7\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:1\n\t"
      "ds_read2_b64 v[72:75], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[76:79], %2 offset0:1\n\t"
      "ds_read2_b64 v[80:83], %3 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[84:87], %3 offset0:1\n\t"
      "ds_read2_b64 v[88:91], %4 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %4 offset0:1\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[96:99], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[100:103], %1 offset0:1\n\t"
      "ds_read2_b64 v[104:107], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[108:111], %2 offset0:1\n\t"
      "ds_read2_b64 v[112:115], %3 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[116:119], %3 offset0:1\n\t"
      "ds_read2_b64 v[120:123], %4 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[124:127], %4 offset0:1\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[64:65], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[72:73], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[80:81], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[88:89], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[66:67], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[74:75], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[82:83], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[90:91], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[68:69], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[76:77], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[84:85], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[92:93], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[70:71], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[78:79], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[86:87], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[94:95], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read2_b64 v[64:67], %1 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[68:71], %1 offset0:1\n\t"
      "ds_read2_b64 v[72:75], %2 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[76:79], %2 offset0:1\n\t"
      "ds_read2_b64 v[80:83], %3 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[84:87], %3 offset0:1\n\t"
      "ds_read2_b64 v[88:91], %4 offset0:3 offset1:2\n\t"
      "ds_read2_b64 v[92:95], %4 offset0:1\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[96:97], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[104:105], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[112:113], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[120:121], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[98:99], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[106:107], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[114:115], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[122:123], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[100:101], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[108:109], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[116:117], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[124:125], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[102:103], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[110:111], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[118:119], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[126:127], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %5, %5, 1\n\t"
      "s_cmp_lg_u32 %5, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a3), "+v"(a2), "+v"(a1), "+v"(a0), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory"
  );
}

/* The region forms with ds_read_b128 (tools/gen_fir_long_asm.py G b128): two adjacent positions of a
 * region per LDS instruction at twice the rate of ds_read2_b64.  Every lane's pair has to sit on a
 * 16-byte boundary: lane stride inside a region even, region size even, lowest position of the batch
 * even -- the two-region window for D = 4 * odd, the four-region window for D = 8 * odd. */
__device__ __forceinline__ void fir_long_e1_b128_asm(fmd_f2v& acc2, unsigned& a1, unsigned& a0, unsigned klo,
                                                     unsigned khi, unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %4\n\t"
      "s_mov_b32 s73, %5\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:48\n\t"
      "ds_read_b128 v[68:71], %1 offset:32\n\t"
      "ds_read_b128 v[72:75], %1 offset:16\n\t"
      "ds_read_b128 v[76:79], %1 offset:0\n\t"
      "ds_read_b128 v[80:83], %2 offset:48\n\t"
      "ds_read_b128 v[84:87], %2 offset:32\n\t"
      "ds_read_b128 v[88:91], %2 offset:16\n\t"
      "ds_read_b128 v[92:95], %2 offset:0\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read_b128 v[96:99], %1 offset:48\n\t"
      "ds_read_b128 v[100:103], %1 offset:32\n\t"
      "ds_read_b128 v[104:107], %1 offset:16\n\t"
      "ds_read_b128 v[108:111], %1 offset:0\n\t"
      "ds_read_b128 v[112:115], %2 offset:48\n\t"
      "ds_read_b128 v[116:119], %2 offset:32\n\t"
      "ds_read_b128 v[120:123], %2 offset:16\n\t"
      "ds_read_b128 v[124:127], %2 offset:0\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[66:67], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[82:83], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[64:65], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[80:81], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[70:71], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[86:87], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[68:69], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[84:85], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[74:75], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[90:91], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[72:73], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[88:89], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[78:79], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[94:95], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[76:77], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[92:93], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:48\n\t"
      "ds_read_b128 v[68:71], %1 offset:32\n\t"
      "ds_read_b128 v[72:75], %1 offset:16\n\t"
      "ds_read_b128 v[76:79], %1 offset:0\n\t"
      "ds_read_b128 v[80:83], %2 offset:48\n\t"
      "ds_read_b128 v[84:87], %2 offset:32\n\t"
      "ds_read_b128 v[88:91], %2 offset:16\n\t"
      "ds_read_b128 v[92:95], %2 offset:0\n\t"
      "v_subrev_u32 %1, 64, %1\n\t"
      "v_subrev_u32 %2, 64, %2\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[98:99], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[114:115], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[96:97], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[112:113], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[102:103], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[118:119], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[100:101], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[116:117], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[106:107], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[122:123], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[104:105], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[120:121], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[110:111], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[126:127], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[108:109], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[124:125], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %3, %3, 1\n\t"
      "s_cmp_lg_u32 %3, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a1), "+v"(a0), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory"
  );
}
__device__ __forceinline__ void fir_long_e2_b128_asm(fmd_f2v& acc2, unsigned& a3, unsigned& a2, unsigned& a1,
                                                     unsigned& a0, unsigned klo, unsigned khi, unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %6\n\t"
      "s_mov_b32 s73, %7\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:16\n\t"
      "ds_read_b128 v[68:71], %1 offset:0\n\t"
      "ds_read_b128 v[72:75], %2 offset:16\n\t"
      "ds_read_b128 v[76:79], %2 offset:0\n\t"
      "ds_read_b128 v[80:83], %3 offset:16\n\t"
      "ds_read_b128 v[84:87], %3 offset:0\n\t"
      "ds_read_b128 v[88:91], %4 offset:16\n\t"
      "ds_read_b128 v[92:95], %4 offset:0\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read_b128 v[96:99], %1 offset:16\n\t"
      "ds_read_b128 v[100:103], %1 offset:0\n\t"
      "ds_read_b128 v[104:107], %2 offset:16\n\t"
      "ds_read_b128 v[108:111], %2 offset:0\n\t"
      "ds_read_b128 v[112:115], %3 offset:16\n\t"
      "ds_read_b128 v[116:119], %3 offset:0\n\t"
      "ds_read_b128 v[120:123], %4 offset:16\n\t"
      "ds_read_b128 v[124:127], %4 offset:0\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[66:67], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[74:75], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[82:83], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[90:91], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[64:65], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[72:73], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[80:81], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[88:89], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[70:71], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[78:79], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[86:87], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[94:95], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[68:69], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[76:77], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[84:85], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[92:93], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:16\n\t"
      "ds_read_b128 v[68:71], %1 offset:0\n\t"
      "ds_read_b128 v[72:75], %2 offset:16\n\t"
      "ds_read_b128 v[76:79], %2 offset:0\n\t"
      "ds_read_b128 v[80:83], %3 offset:16\n\t"
      "ds_read_b128 v[84:87], %3 offset:0\n\t"
      "ds_read_b128 v[88:91], %4 offset:16\n\t"
      "ds_read_b128 v[92:95], %4 offset:0\n\t"
      "v_subrev_u32 %1, 32, %1\n\t"
      "v_subrev_u32 %2, 32, %2\n\t"
      "v_subrev_u32 %3, 32, %3\n\t"
      "v_subrev_u32 %4, 32, %4\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[98:99], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[106:107], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[114:115], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[122:123], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[96:97], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[104:105], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[112:113], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[120:121], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[102:103], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[110:111], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[118:119], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[126:127], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[100:101], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[108:109], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[116:117], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[124:125], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %5, %5, 1\n\t"
      "s_cmp_lg_u32 %5, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a3), "+v"(a2), "+v"(a1), "+v"(a0), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory"
  );
}

/* Long filters in the PLAIN window (E = 0) with D = 2 * odd, two taps per LDS instruction.  With such
 * a window only ONE wave fits a SIMD, and a lone wave issues a packed f32 instruction every 8 cycles
 * whatever depends on what (tools/ubench/pk_rate, profiles/r2_pk_rate.txt): the v_pk_mul_f32 and the
 * v_pk_add_f32 a tap needs without FMA cost 16 cycles, every other instruction comes on top -- the
 * loop is issue-bound, and the fewest instructions per tap win.  ds_read_b128 fetches the samples of
 * taps (j + 1, j) -- consecutive window slots, the lower one 16-byte aligned -- at 256 B/clk/CU where
 * ds_read2_b64 delivers 128 (MI355X_MICROARCH.md, LDS table).  The 16 lanes of a b128 lane group sit
 * 2 D dwords apart = 4 * odd banks (mod 64): 16 different 4-bank groups, conflict-free without
 * de-interleaving.  Batches of 16 taps (8 reads + one s_load_dwordx16) in two register sets,
 * products two ahead of the sum, the sum ONE chain in tap order.  a = LDS byte address of the
 * batch's lowest pair (taps j + 14, j + 15), 128 bytes lower per batch; klo / khi = address of tap
 * j; cnt = pairs of batches (32 taps each, >= 1).  The last batch load is a dummy (16 taps past the
 * table: padded; 16 slots below the last pair: the window sits 32 slots into the LDS allocation).
 * Measured (4096 taps, D = 46): 3.35 ms per launch against 3.51 ms for fir_long_e1_asm on the same
 * box = 19.3 cycles per tap, of which 16 are the two packed instructions.
 * Body generated by tools/gen_fir_long_b128_asm.py. */
__device__ __forceinline__ void fir_long_b128_asm(fmd_f2v& acc2, unsigned& a, unsigned klo, unsigned khi,
                                                  unsigned& cnt)
{
  asm volatile(
      "s_mov_b32 s72, %3\n\t"
      "s_mov_b32 s73, %4\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:112\n\t"
      "ds_read_b128 v[68:71], %1 offset:96\n\t"
      "ds_read_b128 v[72:75], %1 offset:80\n\t"
      "ds_read_b128 v[76:79], %1 offset:64\n\t"
      "ds_read_b128 v[80:83], %1 offset:48\n\t"
      "ds_read_b128 v[84:87], %1 offset:32\n\t"
      "ds_read_b128 v[88:91], %1 offset:16\n\t"
      "ds_read_b128 v[92:95], %1 offset:0\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "1:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[56:71], s[72:73], 0x0\n\t"
      "ds_read_b128 v[96:99], %1 offset:112\n\t"
      "ds_read_b128 v[100:103], %1 offset:96\n\t"
      "ds_read_b128 v[104:107], %1 offset:80\n\t"
      "ds_read_b128 v[108:111], %1 offset:64\n\t"
      "ds_read_b128 v[112:115], %1 offset:48\n\t"
      "ds_read_b128 v[116:119], %1 offset:32\n\t"
      "ds_read_b128 v[120:123], %1 offset:16\n\t"
      "ds_read_b128 v[124:127], %1 offset:0\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[66:67], s[40:41] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[64:65], s[40:41] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[70:71], s[42:43] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[68:69], s[42:43] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[74:75], s[44:45] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[72:73], s[44:45] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[78:79], s[46:47] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[76:77], s[46:47] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[82:83], s[48:49] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[80:81], s[48:49] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[86:87], s[50:51] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[84:85], s[50:51] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[90:91], s[52:53] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[88:89], s[52:53] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[94:95], s[54:55] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[92:93], s[54:55] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[40:55], s[72:73], 0x0\n\t"
      "ds_read_b128 v[64:67], %1 offset:112\n\t"
      "ds_read_b128 v[68:71], %1 offset:96\n\t"
      "ds_read_b128 v[72:75], %1 offset:80\n\t"
      "ds_read_b128 v[76:79], %1 offset:64\n\t"
      "ds_read_b128 v[80:83], %1 offset:48\n\t"
      "ds_read_b128 v[84:87], %1 offset:32\n\t"
      "ds_read_b128 v[88:91], %1 offset:16\n\t"
      "ds_read_b128 v[92:95], %1 offset:0\n\t"
      "v_subrev_u32 %1, 128, %1\n\t"
      "s_add_u32 s72, s72, 64\n\t"
      "s_addc_u32 s73, s73, 0\n\t"
      "v_pk_mul_f32 v[128:129], v[98:99], s[56:57] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 v[130:131], v[96:97], s[56:57] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[102:103], s[58:59] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[100:101], s[58:59] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[106:107], s[60:61] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[104:105], s[60:61] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[110:111], s[62:63] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[108:109], s[62:63] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[114:115], s[64:65] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[112:113], s[64:65] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[118:119], s[66:67] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[116:117], s[66:67] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "v_pk_mul_f32 v[128:129], v[122:123], s[68:69] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "v_pk_mul_f32 v[130:131], v[120:121], s[68:69] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[128:129]\n\t"
      "v_pk_mul_f32 v[132:133], v[126:127], s[70:71] op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, v[130:131]\n\t"
      "v_pk_mul_f32 v[134:135], v[124:125], s[70:71] op_sel:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, v[132:133]\n\t"
      "s_nop 0\n\t"
      "v_pk_add_f32 %0, %0, v[134:135]\n\t"
      "s_sub_u32 %2, %2, 1\n\t"
      "s_cmp_lg_u32 %2, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : "+v"(acc2), "+v"(a), "+s"(cnt)
      : "s"(klo), "s"(khi)
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
        "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134",
        "v135", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64",
        "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "scc", "memory"
  );
}

/* ------------------------------------------------------------------------------------------ */
/* K1: cFineTuner (FmDecode.cpp:66-82) fused into cDownsampleFilter::Process(complex)           */
/*     (DownConvert.cpp:98-154), optionally with the RTL-SDR byte -> float conversion           */
/*     (RTL_SDR_Source.cpp:207-211) in front.  One workgroup = one channel x TILE outputs.      */
/*     The tuned IQ window (overlap-save: (TILE-1)*D + order samples) is staged once in LDS;    */
/*     each thread then accumulates its output over taps j = 1..order in the reference's order. */
/*     Taps are wave-uniform (scalar loads).                                                    */
/* ------------------------------------------------------------------------------------------ */
/* Input formats.  A lane always loads two IQ samples at a time: 16 bytes of complex<float> or 4
 * bytes of RTL-SDR (I,Q) byte pairs, so consecutive lanes write consecutive 16-byte LDS slots. */
struct InF32
{
  typedef float2 elem;
  typedef uint4 pair;
  static __device__ __forceinline__ float2 one(const elem* x, size_t k) { return x[k]; }
  static __device__ __forceinline__ void unpack(const pair& v, float2& a, float2& b)
  {
    a = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    b = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
  }
};
struct InU8
{
  typedef uchar2 elem;
  typedef unsigned pair;
  static __device__ __forceinline__ float2 one(const elem* x, size_t k)
  {
    const uchar2 b = x[k];
    return make_float2(fmd_u8_to_f32(b.x), fmd_u8_to_f32(b.y));
  }
  static __device__ __forceinline__ void unpack(const pair& v, float2& a, float2& b)
  {
    a = make_float2(fmd_u8_to_f32(v & 0xffu), fmd_u8_to_f32((v >> 8) & 0xffu));
    b = make_float2(fmd_u8_to_f32((v >> 16) & 0xffu), fmd_u8_to_f32(v >> 24));
  }
};

/* Staging detail: UNROLL two-sample loads per lane issued back to back (unconditional, index
 * clamped: a branch would make the compiler wait after every load) so a workgroup has its whole
 * window in flight in one round trip.  The tuner table has T | 2*TILE entries (power of two,
 * host-checked), so the two table entries a lane needs are the same for every load it issues and
 * live in registers.
 * Block -> (channel, tile): block ids are dealt round-robin over the 8 XCDs, so with
 * xcd_map != 0 each XCD gets whole channels and walks their tiles in order: consecutive tiles
 * share their `order`-sample halo in that XCD's L2 and every channel is read as one
 * contiguous stream (measured 0.79 ms vs 0.97 ms channel-fastest, 8192 channels).
 * Tried and dropped (round 1, 8192 channels, kernel alone): workgroups that walk several tiles
 * with the next tile's loads prefetched into registers during the tap loop (0.86-0.97 ms vs 0.84
 * on the same box); three outputs per thread sharing their window reads (2.4x less LDS traffic,
 * but 2 waves per SIMD: 0.94-1.09 ms).  Loads + staging alone take 0.84 ms, the tap loop alone
 * 0.66 ms: the float path sits at the HBM rate the chip sustains (6.3 TB/s copy, 79 % of spec). */
/* Window layout, E = log2 of the power-of-two factor of D (template): the lanes of a wave read
 * samples D apart, so with an even D a plain window puts them on a fraction of the LDS banks
 * (D = 46: a half-wave reaches 32 of 64 banks, 2-way conflict; D = 4: 4-way).  The window is
 * therefore stored de-interleaved: slot i (sample k_al + i) lives in region i mod 2^E at position
 * i >> E.  All lanes read the same region at a given tap (their offsets lane*D are multiples of
 * 2^E), and inside a region the lane stride is D >> E, which is odd: conflict-free.  Consecutive
 * taps walk the regions round robin, each region contiguously.  E = 0 is the plain window. */
/* SHFL (opt-in, fmd_params::fir_reduction = 1; plain window, one wave per workgroup only): the tap
 * sum of an output is split over the four lanes of a quad -- lane q takes taps 1+q, 5+q, ... -- and
 * the four partial sums are combined with two wavefront shuffles.  This is the reduction BASELINE's
 * north star describes; it changes the order of the float additions, so its output is NOT
 * bit-identical to the reference's sequential sum (measured against the parity mode in
 * tests/test_gpu_fast_mode.py, figures in DESIGN.md section 3).  Not the default. */
template <class IN, int TILE, int UNROLL, bool POW2, int E = 0, bool LONGASM = false, bool SHFL = false,
          bool RB128 = false>
__global__ __launch_bounds__(TILE) void k_if_fir(const typename IN::elem* __restrict__ iq,
                                                 size_t chan_stride, unsigned N,
                                                 const float2* __restrict__ hist_in,
                                                 float2* __restrict__ hist_out,
                                                 const float2* __restrict__ lut, unsigned T,
                                                 unsigned lut_idx0, const float* __restrict__ coeff,
                                                 unsigned order, unsigned D, unsigned pos, unsigned M,
                                                 float2* __restrict__ out, unsigned Mstride,
                                                 unsigned ntiles, unsigned xcd_map, unsigned cpc)
{
  typedef typename IN::pair pair_t;
  constexpr int G = 1 << E; // regions of the de-interleaved window
  extern __shared__ __attribute__((aligned(16))) float2 smem_win[];
  // the hand-scheduled tap loops' last (dummy) prefetch reaches up to 16 slots below the window (below
  // region 0 in the de-interleaved layouts): keep them inside the allocation
  constexpr int WIN_PAD = LONGASM ? 32 : 0;
  float2* const win = smem_win + WIN_PAD;
  __builtin_amdgcn_s_setprio(1); // ahead of the post-chain kernels it may share a SIMD with
  // region size in slots: the window spans (TILE-1)*D + order samples plus alignment slack
  // (RB128: even, so that every region starts on a 16-byte boundary)
  const unsigned H0 = (((unsigned)(TILE - 1) * D + order + 2u * G + 2u) >> E) + 1u;
  const unsigned H = RB128 ? ((H0 + 1u) & ~1u) : H0;
  auto slot = [&](int i) -> unsigned { // LDS index of window slot i
    return E == 0 ? (unsigned)i : ((unsigned)i & (unsigned)(G - 1)) * H + ((unsigned)i >> E);
  };
  unsigned c, tile;
  if (xcd_map)
  {
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    c = (slot / ntiles) * 8u + xcd;
    tile = slot % ntiles;
  }
  else
  {
    c = blockIdx.x / ntiles;
    tile = blockIdx.x % ntiles;
  }
  const unsigned m0 = tile * TILE;
  const unsigned tid = threadIdx.x;
  const unsigned nout = min((unsigned)TILE, M - m0);
  const int p_first = (int)(pos + m0 * D);
  const int k_lo = p_first - (int)order;            // first sample the tile needs
  const int k_hi = p_first + (int)((nout - 1) * D); // one past the last sample it needs
  // floor to a pair boundary (and to a region-0 slot): window slot of sample k is k - k_al
  const int k_al = k_lo & ~((G > 2 ? G : 2) - 1);
  // (cpc > 1: channels_per_capture consecutive channels tune the same capture, chan_stride apart)
  const typename IN::elem* __restrict__ x = iq + (size_t)(cpc > 1u ? c / cpc : c) * chan_stride;
  const float2* __restrict__ l = lut + (size_t)c * T;

  if (k_lo < 0)
  { // tail of the previous call (already tuned), only for the first tile(s)
    const float2* __restrict__ h = hist_in + (size_t)c * order;
    const int nh = min(-k_lo, k_hi - k_lo);
    for (int i = (int)tid; i < nh; i += TILE)
      win[slot(i + (k_lo - k_al))] = h[(int)order + k_lo + i];
  }
  if (POW2)
  {
    // pair i holds samples k_al + 2i, k_al + 2i + 1; pairs [ifirst, npairs) come from this block,
    // a ragged last sample of an odd-length block is done on its own
    const unsigned mask = T - 1;
    const int kfull = min(k_hi, (int)(N & ~1u));
    const int ifirst = k_al < 0 ? (-k_al) >> 1 : 0;
    const int npairs = (kfull - k_al + 1) >> 1;
    const unsigned li = (lut_idx0 + (unsigned)k_al + 2u * tid) & mask;
    const float2 l0 = l[li], l1 = l[(li + 1) & mask];
    const pair_t* __restrict__ src = reinterpret_cast<const pair_t*>(x) + (k_al >> 1);
    float4* dst = reinterpret_cast<float4*>(win);
    for (int base = 0; base < npairs; base += UNROLL * TILE)
    {
      pair_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        v[u] = src[max(min(base + u * TILE + (int)tid, npairs - 1), ifirst)];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
      {
        const int i = base + u * TILE + (int)tid;
        if (i >= ifirst && i < npairs)
        {
          float2 a, b;
          IN::unpack(v[u], a, b);
          a = cmul(a, l0);
          b = cmul(b, l1);
          if (E == 0)
            dst[i] = make_float4(a.x, a.y, b.x, b.y);
          else
          { // the two samples of a pair belong to neighbouring regions
            win[slot(2 * i)] = a;
            win[slot(2 * i + 1)] = b;
          }
        }
      }
    }
    for (int k = max(kfull, 0) + (int)tid; k < k_hi; k += TILE)
      win[slot(k - k_al)] = cmul(IN::one(x, k), l[(lut_idx0 + (unsigned)k) & mask]);
  }
  else
  {
    for (int k = max(k_al, 0) + (int)tid; k < k_hi; k += TILE)
      win[slot(k - k_al)] = cmul(IN::one(x, k), l[(lut_idx0 + (unsigned)k) % T]);
  }
  // fir_long_b128_asm takes pairs of taps (j, j + 1) whose lower slot -- tap j + 1's -- is even.  The
  // slot of tap j is (k_lo - k_al) + tid * D + order - j; D is even, so its parity is the same for
  // every lane: the pairs start at the first j whose slot is odd, at most one tap goes in front.
  const bool b128 = LONGASM && E == 0 && (D & 3u) == 2u;
  const unsigned jb = (((unsigned)(k_lo - k_al) + order - 1u) & 1u) ? 1u : 2u;
  const unsigned nb128 = (b128 && order + 1u > jb) ? ((order + 1u - jb) >> 5) : 0u; // pairs of batches
  if (TILE == 64)
    lds_wave_sync(); // one wave: its LDS operations execute in order
  else
    __syncthreads();

  if (SHFL && E == 0 && TILE == 64)
  {
    const unsigned q = tid & 3u, og = tid >> 2; // quad lane, output within a pass of 16
#pragma unroll 1
    for (unsigned pass = 0; pass < TILE / 16; pass++)
    {
      const unsigned o = pass * 16 + og;
      float2 acc = make_float2(0.0f, 0.0f);
      if (o < nout)
      {
        const float2* w = win + (k_lo - k_al) + o * D + order;
#pragma unroll 4
        for (unsigned j = 1 + q; j <= order; j += 4)
        {
          const float k = coeff[j]; // per-lane tap: a vector load (L1), not a scalar one
          const float2 s = w[-(int)j];
          acc.x += s.x * k;
          acc.y += s.y * k;
        }
      }
      acc.x += __shfl_xor(acc.x, 1);
      acc.y += __shfl_xor(acc.y, 1);
      acc.x += __shfl_xor(acc.x, 2);
      acc.y += __shfl_xor(acc.y, 2);
      if (q == 0 && o < nout)
        out[(size_t)c * Mstride + m0 + o] = acc;
    }
  }
  else if (tid < nout)
  {
    float2 acc = make_float2(0.0f, 0.0f);
    if (E == 0)
    {
      const float2* w = win + (k_lo - k_al) + tid * D + order; // w[-j] = x[p - j]
      unsigned j = 1;
      if (LONGASM && (D & 1u) && order >= 32u)
      { // odd D: pairs of adjacent slots with ds_read2_b64 (fir_long_odd_asm)
        unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)(order >> 5));
        const unsigned taps = cnt << 5;
        unsigned a0 = (unsigned)(size_t)(w - (int)j - 15); // slot of tap j + 15: the batch's lowest
        const size_t ka = (size_t)(coeff + j);
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
        fmd_f2v acc2 = {acc.x, acc.y};
        fir_long_odd_asm(acc2, a0, klo, khi, cnt);
        acc.x = acc2.x;
        acc.y = acc2.y;
        j += taps;
      }
      else if (b128 && nb128)
      {
        if (jb == 2u)
        { // one tap in front of the pairs
          const float k = coeff[1];
          const float2 s1 = w[-1];
          acc.x += s1.x * k;
          acc.y += s1.y * k;
        }
        j = jb;
        unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)nb128);
        unsigned a = (unsigned)(size_t)(w - (int)j - 15); // slot of tap j + 15: the batch's lowest pair
        const size_t ka = (size_t)(coeff + j);
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
        fmd_f2v acc2 = {acc.x, acc.y};
        fir_long_b128_asm(acc2, a, klo, khi, cnt);
        acc.x = acc2.x;
        acc.y = acc2.y;
        j += nb128 << 5;
      }
#pragma unroll 8
      for (; j <= order; j++)
      {
        const float k = coeff[j];
        const float2 s = w[-(int)j];
        acc.x += s.x * k;
        acc.y += s.y * k;
      }
    }
    else
    {
      // tap j reads window slot U0 - j + tid*D = region (U0 - j) mod G, position ((U0 - j) >> E) +
      // tid * (D >> E): a wave-uniform part plus a per-lane offset
      const unsigned U0 = (unsigned)(k_lo - k_al) + order;
      const float2* lanebase = win + tid * (D >> E);
      unsigned j = 1;
      for (; j <= order && ((U0 - j + 1u) & (unsigned)(G - 1)) != 0u; j++)
      { // until a tap sits in the last region: from there on whole rounds over the regions
        const unsigned u = U0 - j;
        const float2 s = lanebase[(u & (unsigned)(G - 1)) * H + (u >> E)];
        const float k = coeff[j];
        acc.x += s.x * k;
        acc.y += s.y * k;
      }
      if (LONGASM && RB128 && E >= 1 && ((((U0 - j) >> E) & 1u) == 0u) && j + (unsigned)G <= order + 1u)
      { // the 16-byte reads take positions (P - 1, P) with P - 1 even: one round in front when P is even
        const unsigned P = (U0 - j) >> E;
#pragma unroll
        for (int g = 0; g < G; g++)
        {
          const float2 s = lanebase[(unsigned)(G - 1 - g) * H + P];
          const float k = coeff[j + g];
          acc.x += s.x * k;
          acc.y += s.y * k;
        }
        j += (unsigned)G;
      }
      if (LONGASM && RB128 && E >= 1 && order + 1u - j >= 32u)
      {
        unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)((order + 1u - j) >> 5));
        const unsigned taps = cnt << 5;
        const float2* ptop = lanebase + (unsigned)(G - 1) * H + ((U0 - j) >> E) - (16 / G - 1); // region G-1
        const size_t ka = (size_t)(coeff + j);
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
        fmd_f2v acc2 = {acc.x, acc.y};
        if (E == 1)
        {
          unsigned a1 = (unsigned)(size_t)ptop, a0 = (unsigned)(size_t)(ptop - H);
          fir_long_e1_b128_asm(acc2, a1, a0, klo, khi, cnt);
        }
        else
        {
          unsigned a3 = (unsigned)(size_t)ptop, a2 = (unsigned)(size_t)(ptop - H),
                   a1 = (unsigned)(size_t)(ptop - 2u * H), a0 = (unsigned)(size_t)(ptop - 3u * H);
          fir_long_e2_b128_asm(acc2, a3, a2, a1, a0, klo, khi, cnt);
        }
        acc.x = acc2.x;
        acc.y = acc2.y;
        j += taps;
      }
      if (LONGASM && !RB128 && E == 1 && order + 1u - j >= 32u)
      { // see fir_long_e1_asm; whatever is left after whole pairs of batches continues below
        unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)((order + 1u - j) >> 5));
        const unsigned taps = cnt << 5;
        const float2* p1 = lanebase + H + ((U0 - j) >> 1) - 7; // lowest of the batch's 8 positions
        unsigned a1 = (unsigned)(size_t)p1, a0 = (unsigned)(size_t)(p1 - H);
        const size_t ka = (size_t)(coeff + j);
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
        fmd_f2v acc2 = {acc.x, acc.y};
        fir_long_e1_asm(acc2, a1, a0, klo, khi, cnt);
        acc.x = acc2.x;
        acc.y = acc2.y;
        j += taps;
      }
      if (LONGASM && !RB128 && E == 2 && order + 1u - j >= 32u)
      { // see fir_long_e2_asm: tap j is in region 3 here, four positions per region and batch
        unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)((order + 1u - j) >> 5));
        const unsigned taps = cnt << 5;
        const float2* p3 = lanebase + 3u * H + ((U0 - j) >> 2) - 3; // lowest of the batch's 4 positions
        unsigned a3 = (unsigned)(size_t)p3, a2 = (unsigned)(size_t)(p3 - H), a1 = (unsigned)(size_t)(p3 - 2u * H),
                 a0 = (unsigned)(size_t)(p3 - 3u * H);
        const size_t ka = (size_t)(coeff + j);
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
        const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
        fmd_f2v acc2 = {acc.x, acc.y};
        fir_long_e2_asm(acc2, a3, a2, a1, a0, klo, khi, cnt);
        acc.x = acc2.x;
        acc.y = acc2.y;
        j += taps;
      }
      // whole rounds: tap j + g sits in region G-1-g at position ((U0 - j) >> E), one lower per round
      const float* __restrict__ kp = coeff + j;
      const float2* p = lanebase + ((U0 - j) >> E);
      const unsigned nrounds = (order + 1u - j) >> E;
      constexpr int ROUNDS = 8 >> E; // 8 taps per unrolled body, as in the plain loop
#pragma unroll ROUNDS
      for (unsigned r = 0; r < nrounds; r++)
      {
#pragma unroll
        for (int g = 0; g < G; g++)
        {
          const float2 s = p[(ptrdiff_t)((unsigned)(G - 1 - g) * H) - (ptrdiff_t)r];
          const float k = kp[r * G + g];
          acc.x += s.x * k;
          acc.y += s.y * k;
        }
      }
      j += nrounds << E;
      for (; j <= order; j++)
      {
        const unsigned u = U0 - j;
        const float2 s = lanebase[(u & (unsigned)(G - 1)) * H + (u >> E)];
        const float k = coeff[j];
        acc.x += s.x * k;
        acc.y += s.y * k;
      }
    }
    out[(size_t)c * Mstride + m0 + tid] = acc;
  }

  // the workgroup of the last tile also saves the last `order` tuned samples (:135-151); a block
  // shorter than the filter keeps the newest part of the old history in front of it (:137-145)
  if (tile == ntiles - 1)
  {
    float2* __restrict__ ho = hist_out + (size_t)c * order;
    const float2* __restrict__ hi = hist_in + (size_t)c * order;
    const unsigned keep = N < order ? order - N : 0u;
    for (unsigned i = tid; i < order; i += TILE)
    {
      if (i < keep)
        ho[i] = hi[i + N];
      else
      {
        const unsigned k = N + i - order;
        ho[i] = cmul(IN::one(x, k), l[(lut_idx0 + k) % T]);
      }
    }
  }
}

/* The headline geometry (plain window, one wave per workgroup, power-of-two tuner table) with NT
 * consecutive tiles of one channel per workgroup: the loads of tile t+1 are issued, into registers,
 * before the tap loop of tile t.  A CU's LDS (25 windows) is all the data k_if_fir keeps in flight,
 * and only while a workgroup waits for its loads; here a wave always has a tile's loads in flight
 * (in its registers) next to the tile it computes on (in LDS).  Same arithmetic, same order. */
template <class IN, int UNROLL, int NT>
__global__ __launch_bounds__(64) void k_if_fir_mt(const typename IN::elem* __restrict__ iq,
                                                  size_t chan_stride, unsigned N,
                                                  const float2* __restrict__ hist_in,
                                                  float2* __restrict__ hist_out,
                                                  const float2* __restrict__ lut, unsigned T,
                                                  unsigned lut_idx0, const float* __restrict__ coeff,
                                                  unsigned order, unsigned D, unsigned pos, unsigned M,
                                                  float2* __restrict__ out, unsigned Mstride,
                                                  unsigned ntiles, unsigned xcd_map, unsigned cpc)
{
  typedef typename IN::pair pair_t;
  constexpr int TILE = 64;
  extern __shared__ __attribute__((aligned(16))) float2 win[];
  __builtin_amdgcn_s_setprio(1);
  const unsigned ngroups = (ntiles + NT - 1) / NT;
  unsigned c, tg;
  if (xcd_map)
  {
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    c = (slot / ngroups) * 8u + xcd;
    tg = slot % ngroups;
  }
  else
  {
    c = blockIdx.x / ngroups;
    tg = blockIdx.x % ngroups;
  }
  const unsigned tid = threadIdx.x;
  // (cpc > 1: channels_per_capture consecutive channels tune the same capture, chan_stride apart)
  const typename IN::elem* __restrict__ x = iq + (size_t)(cpc > 1u ? c / cpc : c) * chan_stride;
  const float2* __restrict__ l = lut + (size_t)c * T;
  const unsigned mask = T - 1;
  const int kfull_all = (int)(N & ~1u);
  // the two table entries of a lane are the same for every tile (T | TILE * D, host-checked)
  float2 l0, l1;
  {
    const int k_al0 = ((int)(pos + tg * NT * TILE * D) - (int)order) & ~1;
    const unsigned li = (lut_idx0 + (unsigned)k_al0 + 2u * tid) & mask;
    l0 = l[li];
    l1 = l[(li + 1) & mask];
  }
  pair_t v[UNROLL];
  auto issue = [&](unsigned tile) { // the tile's window, two samples per lane and load
    const unsigned m0 = tile * TILE;
    const unsigned nout = min((unsigned)TILE, M - m0);
    const int p_first = (int)(pos + m0 * D);
    const int k_lo = p_first - (int)order;
    const int k_hi = p_first + (int)((nout - 1) * D);
    const int k_al = k_lo & ~1;
    const int kfull = min(k_hi, kfull_all);
    const int ifirst = k_al < 0 ? (-k_al) >> 1 : 0;
    const int npairs = (kfull - k_al + 1) >> 1;
    const pair_t* __restrict__ src = reinterpret_cast<const pair_t*>(x) + (k_al >> 1);
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      v[u] = src[max(min(u * TILE + (int)tid, npairs - 1), ifirst)];
  };
  const unsigned t_first = tg * NT;
  issue(t_first);
  for (unsigned i = 0; i < (unsigned)NT; i++)
  {
    const unsigned tile = t_first + i;
    if (tile >= ntiles)
      break;
    const unsigned m0 = tile * TILE;
    const unsigned nout = min((unsigned)TILE, M - m0);
    const int p_first = (int)(pos + m0 * D);
    const int k_lo = p_first - (int)order;
    const int k_hi = p_first + (int)((nout - 1) * D);
    const int k_al = k_lo & ~1;
    if (k_lo < 0)
    { // tail of the previous call (already tuned), only for the first tile(s)
      const float2* __restrict__ h = hist_in + (size_t)c * order;
      const int nh = min(-k_lo, k_hi - k_lo);
      for (int q = (int)tid; q < nh; q += TILE)
        win[q + (k_lo - k_al)] = h[(int)order + k_lo + q];
    }
    {
      const int kfull = min(k_hi, kfull_all);
      const int ifirst = k_al < 0 ? (-k_al) >> 1 : 0;
      const int npairs = (kfull - k_al + 1) >> 1;
      float4* dst = reinterpret_cast<float4*>(win);
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
      {
        const int q = u * TILE + (int)tid;
        if (q >= ifirst && q < npairs)
        {
          float2 a, b;
          IN::unpack(v[u], a, b);
          a = cmul(a, l0);
          b = cmul(b, l1);
          dst[q] = make_float4(a.x, a.y, b.x, b.y);
        }
      }
      // pairs beyond one round of loads (never with the UNROLL the host picks) and a ragged last sample
      const pair_t* __restrict__ src = reinterpret_cast<const pair_t*>(x) + (k_al >> 1);
      for (int q = UNROLL * TILE + (int)tid; q < npairs; q += TILE)
      {
        float2 a, b;
        IN::unpack(src[max(q, ifirst)], a, b);
        a = cmul(a, l0);
        b = cmul(b, l1);
        if (q >= ifirst)
          dst[q] = make_float4(a.x, a.y, b.x, b.y);
      }
      for (int k = max(kfull, 0) + (int)tid; k < k_hi; k += TILE)
        win[k - k_al] = cmul(IN::one(x, k), l[(lut_idx0 + (unsigned)k) & mask]);
    }
    lds_wave_sync();
    if (i + 1 < (unsigned)NT && tile + 1 < ntiles)
      issue(tile + 1); // in flight during the tap loop below
    if (tid < nout)
    {
      float2 acc = make_float2(0.0f, 0.0f);
      const float2* w = win + (k_lo - k_al) + tid * D + order; // w[-j] = x[p - j]
#pragma unroll 8
      for (unsigned j = 1; j <= order; j++)
      {
        const float k = coeff[j];
        const float2 sm = w[-(int)j];
        acc.x += sm.x * k;
        acc.y += sm.y * k;
      }
      out[(size_t)c * Mstride + m0 + tid] = acc;
    }
    if (tile == ntiles - 1)
    {
      float2* __restrict__ ho = hist_out + (size_t)c * order;
      const float2* __restrict__ hi = hist_in + (size_t)c * order;
      const unsigned keep = N < order ? order - N : 0u;
      for (unsigned q = tid; q < order; q += TILE)
      {
        if (q < keep)
          ho[q] = hi[q + N];
        else
        {
          const unsigned k = N + q - order;
          ho[q] = cmul(IN::one(x, k), l[(lut_idx0 + k) % T]);
        }
      }
    }
    lds_wave_sync(); // this tile's window reads are done before the next tile's staging
  }
}

/* k_if_fir_mt with RO = 2 or 3 adjacent outputs per lane (a wave = 128 / 192 outputs).  The pipeline runs the
 * package at its power limit, and what the IF FIR's tap loop burns sets the clock of everything else (with half
 * the taps -- an experiment -- the serial stage beside it takes 1.55 instead of 1.76 ms): the outputs m, m + 1
 * (, m + 2) of a lane share all but D (2 D) of their window, so every tuned sample is read from LDS once for up
 * to RO taps (99 reads for two outputs, 110 for three, instead of 88 each at D = 11; lane stride RO D samples:
 * 66 words for three, conflict-free; 44 for two, two-way conflicts on half as many reads).  Each output still
 * adds its taps j = 1 .. order in the reference's order; a sample at distance o below the newest output's
 * position is tap o - (RO - 1 - r) D of output r.  Two per lane is what runs: 13 waves per CU instead of 9,
 * the FIR 0.975 instead of 1.00 ms inside the pipeline, the whole path the same or better (fmd_batch.hip). */
/* if_level != nullptr: RMSLevelApprox + its moving average (FmDecode.cpp:427, 505-519; k_if_level) in the
 * workgroup that owns a channel's first tile -- the tuned samples 0 .. (N + 63) / 64 - 1 it measures are in that
 * tile's window already (the window starts `order` samples before the first output position, which is < D).  The
 * lanes form the |s|^2 terms, sixteen each, lane 0 adds them in index order like the reference's loop: ~3 us in
 * one workgroup of 23 per channel, against a launch of its own that re-read and re-tuned the samples (0.07 GB per
 * call at 8192 channels). */
template <class IN, int UNROLL, int NT, int RO = 3, int ORD = 88, int DEC = 11>
__global__ __launch_bounds__(64) void k_if_fir_mt3(const typename IN::elem* __restrict__ iq,
                                                   size_t chan_stride, unsigned N,
                                                   const float2* __restrict__ hist_in,
                                                   float2* __restrict__ hist_out,
                                                   const float2* __restrict__ lut, unsigned T,
                                                   unsigned lut_idx0, const float* __restrict__ coeff,
                                                   unsigned order, unsigned D, unsigned pos, unsigned M,
                                                   float2* __restrict__ out, unsigned Mstride,
                                                   unsigned ntiles, unsigned xcd_map, unsigned cpc,
                                                   float* __restrict__ if_level)
{
  typedef typename IN::pair pair_t;
  // (RO = 5, conflict-free too, 132 reads for five outputs: 267 VGPRs and 29 KB of window, one wave per
  // SIMD -- the FIR takes 1.40 ms inside the pipeline, the serial stage beside it its 1.42 alone)
  static_assert(RO == 2 || RO == 3, "outputs per lane (lane stride RO D samples: conflict-free for 3)");
  constexpr int LANES = 64, TILE = LANES * RO;
  extern __shared__ __attribute__((aligned(16))) float2 win[];
  __builtin_amdgcn_s_setprio(1);
  const unsigned ngroups = (ntiles + NT - 1) / NT;
  unsigned c, tg;
  if (xcd_map)
  {
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    c = (slot / ngroups) * 8u + xcd;
    tg = slot % ngroups;
  }
  else
  {
    c = blockIdx.x / ngroups;
    tg = blockIdx.x % ngroups;
  }
  const unsigned tid = threadIdx.x;
  // (cpc > 1: channels_per_capture consecutive channels tune the same capture, chan_stride apart)
  const typename IN::elem* __restrict__ x = iq + (size_t)(cpc > 1u ? c / cpc : c) * chan_stride;
  const float2* __restrict__ l = lut + (size_t)c * T;
  const unsigned mask = T - 1;
  const int kfull_all = (int)(N & ~1u);
  // the two table entries of a lane are the same for every load and tile (T | 128 and T | TILE * D, host-checked)
  float2 l0, l1;
  {
    const int k_al0 = ((int)(pos + tg * NT * TILE * D) - (int)order) & ~1;
    const unsigned li = (lut_idx0 + (unsigned)k_al0 + 2u * tid) & mask;
    l0 = l[li];
    l1 = l[(li + 1) & mask];
  }
  pair_t v[UNROLL];
  auto issue = [&](unsigned tile) { // the tile's window, two samples per lane and load
    const unsigned m0 = tile * TILE;
    const unsigned nout = min((unsigned)TILE, M - m0);
    const int p_first = (int)(pos + m0 * D);
    const int k_lo = p_first - (int)order;
    const int k_hi = p_first + (int)((nout - 1) * D);
    const int k_al = k_lo & ~1;
    const int kfull = min(k_hi, kfull_all);
    const int ifirst = k_al < 0 ? (-k_al) >> 1 : 0;
    const int npairs = (kfull - k_al + 1) >> 1;
    const pair_t* __restrict__ src = reinterpret_cast<const pair_t*>(x) + (k_al >> 1);
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      v[u] = src[max(min(u * LANES + (int)tid, npairs - 1), ifirst)];
  };
  const unsigned t_first = tg * NT;
  issue(t_first);
  for (unsigned i = 0; i < (unsigned)NT; i++)
  {
    const unsigned tile = t_first + i;
    if (tile >= ntiles)
      break;
    const unsigned m0 = tile * TILE;
    const unsigned nout = min((unsigned)TILE, M - m0);
    const int p_first = (int)(pos + m0 * D);
    const int k_lo = p_first - (int)order;
    const int k_hi = p_first + (int)((nout - 1) * D);
    const int k_al = k_lo & ~1;
    if (k_lo < 0)
    { // tail of the previous call (already tuned), only for the first tile(s)
      const float2* __restrict__ h = hist_in + (size_t)c * order;
      const int nh = min(-k_lo, k_hi - k_lo);
      for (int q = (int)tid; q < nh; q += LANES)
        win[q + (k_lo - k_al)] = h[(int)order + k_lo + q];
    }
    {
      const int kfull = min(k_hi, kfull_all);
      const int ifirst = k_al < 0 ? (-k_al) >> 1 : 0;
      const int npairs = (kfull - k_al + 1) >> 1;
      float4* dst = reinterpret_cast<float4*>(win);
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
      {
        const int q = u * LANES + (int)tid;
        if (q >= ifirst && q < npairs)
        {
          float2 a, b;
          IN::unpack(v[u], a, b);
          a = cmul(a, l0);
          b = cmul(b, l1);
          dst[q] = make_float4(a.x, a.y, b.x, b.y);
        }
      }
      // pairs beyond one round of loads (never with the UNROLL the host picks) and a ragged last sample
      const pair_t* __restrict__ src = reinterpret_cast<const pair_t*>(x) + (k_al >> 1);
      for (int q = UNROLL * LANES + (int)tid; q < npairs; q += LANES)
      {
        float2 a, b;
        IN::unpack(src[max(q, ifirst)], a, b);
        a = cmul(a, l0);
        b = cmul(b, l1);
        if (q >= ifirst)
          dst[q] = make_float4(a.x, a.y, b.x, b.y);
      }
      for (int k = max(kfull, 0) + (int)tid; k < k_hi; k += LANES)
        win[k - k_al] = cmul(IN::one(x, k), l[(lut_idx0 + (unsigned)k) & mask]);
    }
    lds_wave_sync();
    if (i + 1 < (unsigned)NT && tile + 1 < ntiles)
      issue(tile + 1); // in flight during the tap loop below
    if (tid * RO < nout)
    {
      // (order == ORD and D == DEC, host-checked, ORD a multiple of DEC: the window goes in stretches of DEC
      // samples, each unrolled; with run-time bounds the compiler's remainder loops wait for a scalar load
      // and an LDS read per sample, unrolled as a whole it holds all 110 samples in registers)
      static_assert(ORD % DEC == 0 && ORD >= RO * DEC, "stretches of DEC samples");
      constexpr int NB = ORD / DEC, NS = NB + (RO - 1); // stretches of an output's taps / of the lane's window
      float2 acc[RO]; // acc[r]: output r of the lane, RO - 1 the newest
#pragma unroll
      for (int r = 0; r < RO; r++)
        acc[r] = make_float2(0.0f, 0.0f);
      // w[-o] = the sample o below the newest output's position: tap o - (RO - 1 - r) D of output r
      const float2* w = win + (k_lo - k_al) + (tid * RO + (RO - 1)) * DEC + ORD;
      // stretch st: output r takes its taps (st - (RO - 1 - r)) D + 1 ... + D, if that is one of its NB stretches
      auto stretch = [&](int st, auto all) {
        const float2* ws = w - st * DEC;
#pragma unroll
        for (int i = 1; i <= DEC; i++)
        {
          const float2 sm = ws[-i];
#pragma unroll
          for (int r = 0; r < RO; r++)
          {
            const int sb = st - (RO - 1 - r);
            if (decltype(all)::value || (sb >= 0 && sb < NB))
            {
              const float cj = coeff[sb * DEC + i];
              acc[r].x += sm.x * cj;
              acc[r].y += sm.y * cj;
            }
          }
        }
      };
#pragma unroll
      for (int st = 0; st < RO - 1; st++) // the newest outputs only
        stretch(st, std::false_type{});
#pragma unroll 1
      for (int st = RO - 1; st < NB; st++)
        stretch(st, std::true_type{});
#pragma unroll
      for (int st = NB; st < NS; st++) // the oldest outputs only
        stretch(st, std::false_type{});
      float2* __restrict__ op = out + (size_t)c * Mstride + m0 + tid * RO;
#pragma unroll
      for (int r = 0; r < RO; r++)
        if (tid * RO + r < nout)
          op[r] = acc[r];
    }
    if (if_level != nullptr && tile == 0)
    { // the level meter over the first n tuned samples (all inside this window: n <= 1024 <= k_hi, host-checked)
      const unsigned n = (N + 63u) / 64u;
      const float2* __restrict__ w0 = win - k_al; // w0[k] = tuned sample k
      float term[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
      {
        const unsigned kk = min(tid * 16u + (unsigned)u, n - 1u);
        const float2 sm = w0[kk];
        term[u] = sm.x * sm.x + sm.y * sm.y; // re * re + im * im (FmDecode.cpp:514)
      }
      lds_wave_sync(); // every lane has read its samples: the terms take the window's place
      float4* __restrict__ tw = reinterpret_cast<float4*>(win);
#pragma unroll
      for (int u = 0; u < 4; u++)
        tw[tid * 4u + (unsigned)u] = make_float4(term[4 * u], term[4 * u + 1], term[4 * u + 2], term[4 * u + 3]);
      lds_wave_sync();
      if (tid == 0)
      {
        const float* __restrict__ tf = reinterpret_cast<const float*>(win);
        float level = 0.0f;
        for (unsigned q = 0; q < n; ++q)
          level += tf[q];
        const float rms = sqrtf(level / (float)n);
        if_level[c] = 0.95f * if_level[c] + 0.05f * rms;
      }
    }
    if (tile == ntiles - 1)
    {
      float2* __restrict__ ho = hist_out + (size_t)c * order;
      const float2* __restrict__ hi = hist_in + (size_t)c * order;
      const unsigned keep = N < order ? order - N : 0u;
      for (unsigned q = tid; q < order; q += LANES)
      {
        if (q < keep)
          ho[q] = hi[q + N];
        else
        {
          const unsigned k = N + q - order;
          ho[q] = cmul(IN::one(x, k), l[(lut_idx0 + k) % T]);
        }
      }
    }
    lds_wave_sync(); // this tile's window reads are done before the next tile's staging
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K2a: RMSLevelApprox (FmDecode.cpp:505-519) + EMA (:427).  One wave per channel: the lanes    */
/*      form the |tuned sample|^2 terms (coalesced), lane 0 adds them in index order.           */
/* ------------------------------------------------------------------------------------------ */
template <class IN>
__global__ __launch_bounds__(64) void k_if_level(const typename IN::elem* __restrict__ iq,
                                                 size_t chan_stride, unsigned N,
                                                 const float2* __restrict__ lut, unsigned T,
                                                 unsigned lut_idx0, ChannelState st, unsigned cpc)
{
  __shared__ float term[1024];
  const unsigned c = blockIdx.x;
  const unsigned n = (N + 63) / 64; // <= 1024 for N <= 65536
  // (cpc > 1: channels_per_capture consecutive channels tune the same capture, chan_stride apart)
  const typename IN::elem* __restrict__ x = iq + (size_t)(cpc > 1u ? c / cpc : c) * chan_stride;
  const float2* __restrict__ l = lut + (size_t)c * T;
  for (unsigned i = threadIdx.x; i < n; i += 64)
  {
    const float2 s = cmul(IN::one(x, i), l[(lut_idx0 + i) % T]);
    term[i] = s.x * s.x + s.y * s.y;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    float level = 0.0f;
    for (unsigned i = 0; i < n; ++i)
      level += term[i];
    const float rms = sqrtf(level / (float)n);
    st.F(F_IF_LEVEL)[c] = 0.95f * st.F(F_IF_LEVEL)[c] + 0.05f * rms;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K2: everything that is a sample-by-sample recurrence at the baseband rate, one lane per     */
/*     channel, 64 channels per workgroup.  The workgroup has TWO waves with different roles   */
/*     (they sit on different SIMDs of the CU, so they issue in parallel):                      */
/*       wave 0: FM PLL recurrence (FmDecode.cpp:362-408) -> NCO frequency term chunk in LDS    */
/*       wave 1: the PLL's output filter (:409-412), SamplesMeanRMS (:522-539),                 */
/*               cPilotPhaseLock::Process (:143-229) with the                                   */
/*               2*baseband multiply (:455-456), RDS quadrature-oscillator mix                  */
/*               (DownConvert.cpp:429-466), and all stores                                      */
/*     Chunks of DS samples are double-buffered in LDS, one barrier per chunk.  A lone wave     */
/*     issues one VALU op every ~4 cycles, so the longest role sets the time per sample.       */
/* ------------------------------------------------------------------------------------------ */

/* NG = channel groups (of 64) per workgroup, one pair of role waves each; NG = 2 puts one role wave
 * on each SIMD of a CU.  EXCL: every wave claims the whole register file of its SIMD (512 = 256 arch +
 * 256 acc VGPRs), so that no bandwidth kernel's wave shares a SIMD with a role wave (such a wave
 * delays the recurrence's instructions by up to one 4-cycle issue each).  That was the default while
 * the stage was the period of the pipeline; now that it has slack the launch leaves the claim off
 * (fmd_batch.hip). */
/* Measured and dropped: four groups per workgroup with the two role waves of a group on ONE
 * SIMD (half the register file each, 32 CUs owned): the waves do not fit into each other's issue
 * gaps, the stage takes 3.5 ms (151 GS/s). */
#ifndef FMD_DS
#define FMD_DS 32
#endif
constexpr int DS = FMD_DS; // samples per LDS chunk (32 or 16)
static_assert(DS == 32 || DS == 16, "chunk size");
typedef float fmd_v4f __attribute__((ext_vector_type(4)));
constexpr int STAGE_RS = 65; // row stride of the staged input in LDS (float2 units)
constexpr unsigned FM_UNROLL = 4; // samples per trip of the FM wave's loop over a full chunk (1, 2, 4: 651 / 599 / 596 cycles per sample)

/* MIX = false (large batches, where k_halfband_chain follows): the stage neither runs the RDS oscillator nor
 * writes the mixed rows.  The oscillator (DownConvert.cpp:436-442) is a recurrence on its own state only
 * -- the same numbers for every channel of a batch (the host computes them once per call, rds_osc_table
 * in fmd_batch.hip) -- and the
 * product with the baseband is made where it is consumed (k_halfband_chain<.., true>): one store per
 * sample instead of two, ~20 instructions per sample less in the second role wave, 0.39 GB per call less
 * (8192 channels).  `osc_after_*` = the oscillator state behind this call: the per-channel copy that the
 * MIX = true form keeps in registers is brought up to date from it. */
template <int NG, bool EXCL, bool MIX = true>
__global__ __launch_bounds__(128 * NG) void k_demod_serial(
    const float2* __restrict__ demod, unsigned Mstride, unsigned M, unsigned C, unsigned CP,
    DemodConsts k, ChannelState st, float2* __restrict__ br, unsigned Hbb,
    float2* __restrict__ mix, unsigned Hmix, const double* __restrict__ sctab_g, FmdSincosTab sct,
    unsigned stereo_q, long long* __restrict__ wg_probe, float osc_after_re, float osc_after_im)
{
  // sctab_g: (sin, cos)(k / 256), 2048 entries (fmd_sincos_p256)
  // dev aid ("serial_probe" of fmd_batch_debug_set): when each workgroup started and ended on the
  // 100 MHz clock, and its shader-clock cycles in between
  /* LDS per workgroup: tables 32 KB + 2.5 KB, per group chunk 16 KB + staged input 33 KB: 84 KB with one
   * group (NG = 1, the shared form above 8192 channels and in serialised mode), 133 KB with two.  84 KB
   * is more than half a CU's 160 KB: ONE one-group workgroup (2 waves) per CU, so 32 768 channels = 512
   * workgroups take two rounds on 256 CUs.  Measured at 32 768 channels (profiles/r*_bench_32768ch.json):
   * the batch is throughput-bound by the bandwidth kernels there and shows no loss against the 66 KB of
   * round 2 (1024-entry table, rows of 64), which did fit twice; whoever grows this further should look. */
  const long long probe_r0 = wg_probe ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  const long long probe_c0 = wg_probe ? (long long)__builtin_readcyclecounter() : 0;
  __shared__ float chunk_all[NG][2][DS][64];  // baseband, FM role -> pilot/RDS role
  // IF-FIR output, pilot/RDS role -> FM role.  Rows of 65: the staging writes a lane's two samples
  // of one channel, lanes 16 apart in rows 2 apart -- with rows of 64 that is one bank pair for 16 lanes
  __shared__ float2 stage_all[NG][2][DS][STAGE_RS];
  // the larger alignment puts the tables first in the LDS layout: below 64 KB their base folds
  // into the read's offset field (one instruction less on the path from the phase to its sine)
  constexpr unsigned SCTAB_N = FMD_SINCOS_P256_SIZE;
  __shared__ __attribute__((aligned(1024))) double sctab[2 * SCTAB_N];
  __shared__ __attribute__((aligned(512))) float atab[FMD_ATAN_TAB_FLOATS];
  /* Chunk hand-off between the two role waves of a group.  One group per workgroup: a barrier per
   * chunk.  Several groups: a barrier would also make the groups wait for each other every chunk
   * (measured +3.8 % cycles with two groups), so each pair keeps two progress counters instead:
   * done[g][0] = chunks the FM wave has written, done[g][1] = iterations the second wave has
   * finished (= chunks it has staged ahead). */
  constexpr bool PAIRSYNC = NG > 1;
  __shared__ unsigned done_all[NG][2];
  if (threadIdx.x < 2 * NG)
    (&done_all[0][0])[threadIdx.x] = 0;
  // latency-bound recurrence: when bandwidth kernels of other calls share the SIMD, issue first
  __builtin_amdgcn_s_setprio(3);
  if (EXCL)
    asm volatile("" ::: "v255", "a255");
  for (unsigned i = threadIdx.x; i < 2 * SCTAB_N; i += 128 * NG)
    sctab[i] = sctab_g[i];
  if (threadIdx.x == 0)
    fmd_atan_table_fill(atab);
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = threadIdx.x >> 6;
  const unsigned role = wave & 1u;
  const unsigned grp = wave >> 1;
  float (*chunk)[DS][64] = chunk_all[grp];
  float2 (*stage)[DS][STAGE_RS] = stage_all[grp];
  const unsigned done_fm = (unsigned)(size_t)&done_all[grp][0];  // LDS byte addresses
  const unsigned done_2nd = (unsigned)(size_t)&done_all[grp][1];
  // a constant of the sine series, pinned in a vector register for both sample loops
  double m16 = -1.0 / 6.0;
  asm volatile("" : "+v"(m16));
  const unsigned c0 = (blockIdx.x * NG + grp) * 64 + lane;
  const bool active = c0 < C;
  const unsigned c = active ? c0 : C - 1; // padded lanes shadow the last channel, stores masked
  const unsigned nchunks = (M + DS - 1) / DS;
  const float2* __restrict__ row = demod + (size_t)c * Mstride;
  // chunk 0 of the input: both waves fetch half of it
  for (unsigned u = role; u < DS; u += 2)
    stage[0][u][lane] = row[min(u, M - 1)];
  __syncthreads();

  if (role == 0)
  {
    float nco_phase = st.F(F_NCO_PHASE)[c], nco_incr = st.F(F_NCO_INCR)[c];
    __builtin_amdgcn_s_waitcnt(0); // state in registers: no memory wait is left inside the loop
    for (unsigned j = 0; j <= nchunks; j++)
    {
      if (PAIRSYNC)
      { // stage[j & 1] staged and chunk[j & 1] read: the second wave has finished iteration j - 1
        if (j == nchunks)
          break;
        lds_wait_ge(done_2nd, j, st.spin_limit, st.err);
      }
      if (j < nchunks)
      {
        const unsigned m0 = j * DS;
        const unsigned cnt = min((unsigned)DS, M - m0);
        /* One sample of the FM PLL (FmDecode.cpp:371-413).  The wave is bound by the number of
         * instructions it issues (one wave per SIMD, ~5 cycles each whatever their class): everything
         * below is written for that count.  Returns whether the sample met a rare input (arctangent
         * outside the table form's range): its result is then meaningless and the caller redoes it. */
        auto fm_sample = [&](unsigned u) -> uint32_t {
          const float2 sin_ = stage[j & 1][u][lane]; // staged one chunk ahead by the other wave
          const float sre = sin_.x, sim = sin_.y;
          float sn, cs;
          fmd_sincos_p256_finish(fmd_sincos_p256_lookup_lds(nco_phase, sctab), m16, &sn, &cs);
          // ComplexType(Cos, Sin) * signal[i] as three packed operations:
          // (cs sre, cs sim) + (-(sn sim), sn sre)  [fmd_pk_add_cross: (a.x - b.y, a.y + b.x)]
          const fmd_v2f dd = fmd_pk_add_cross((fmd_v2f){sre, sim} * cs, (fmd_v2f){sre, sim} * sn);
          const float dre = dd.x, dim = dd.y;
          uint32_t lit; // >= FMD_ATAN_RARE_LIMIT: the sample needs the literal path
          const float err = -fmd_atan2f_tab_core(dim, dre, atab, &lit);
          /* :399-402 as max / min: the same as the reference's two compares for every number; a
           * NaN state (only ever out of non-finite input) goes through the literal path */
          const fmd_v2f ba = (fmd_v2f){k.pll_beta, k.pll_alpha} * err;
          nco_incr += ba.x;
          nco_incr = fminf(fmaxf(nco_incr, k.nco_ll), k.nco_hl);
          nco_phase += nco_incr + ba.y;
          {
            /* :404-407  if (phase >= 2pi) phase = fmod(phase, 2pi); while (phase < 0) phase += 2pi;
             * For phase in [2pi, 4pi) fmod is the exact difference phase - 2pi, and for
             * [-2pi, 0) the loop runs once.  The new phase cannot be outside (-2 pi, 4 pi): the old one
             * lies in [0, 2 pi] (by this very wrap), the increment is clamped to +-0.95 pi and
             * alpha |err| <= 0.67 pi.  A NaN anywhere (only ever out of non-finite input) makes the
             * quotient inside the arctangent NaN, i.e. `lit`. */
            /* K_2PI lies between the floats 0x40c90fda and 0x40c90fdb, so "0 <= phase < K_2PI" is one
             * unsigned compare of the float's bits (the phase is never -0: a sum is -0 only out of two
             * -0, the state starts at +0 and a wrapped phase is never 0 at all).  The offset -2 pi / 0 /
             * +2 pi is built as a double's high word -- sign = the phase's inverted, everything else K_2PI's,
             * or all zero in range -- over K_2PI's low word: in range that is a subnormal (below
             * 2^-1043) and phase + it rounds back to the phase, so no select of the result is needed. */
            const uint32_t pb = fmd_f2u(nco_phase);
            uint32_t khi = (~pb & 0x80000000u) | 0x401921fbu;
            khi = pb < 0x40c90fdbu ? 0u : khi;
            const uint64_t kb = ((uint64_t)khi << 32) | 0x54442d18u;
            double off;
            memcpy(&off, &kb, 8);
            nco_phase = (float)((double)nco_phase + off); // exact difference / sum, rounded once
          }
          // the NCO increment; phaseIncr = 2 * increment (:409) and the output filter run in wave 1
          chunk[j & 1][u][lane] = nco_incr;
          return lit;
        };
        /* The same sample written out literally (fdlibm arctangent as glibc has it, the reference's
         * compares and its fmod): what a rare input gets, and -- identical for every other input --
         * what the rest of its group is redone with. */
        auto fm_sample_literal = [&](unsigned u) {
          const float2 sin_ = stage[j & 1][u][lane];
          float sn, cs;
          fmd_sincos_p256k(nco_phase, sctab, m16, &sn, &cs);
          const fmd_v2f dd = fmd_pk_add_cross((fmd_v2f){sin_.x, sin_.y} * cs, (fmd_v2f){sin_.x, sin_.y} * sn);
          const float e2 = -fmd_atan2f(dd.y, dd.x);
          float in2 = nco_incr + k.pll_beta * e2;
          in2 = (in2 < k.nco_ll) ? k.nco_ll : in2;
          in2 = (in2 > k.nco_hl) ? k.nco_hl : in2;
          float ph2 = nco_phase + (in2 + k.pll_alpha * e2);
          const double pd2 = (double)ph2;
          if (pd2 >= FMD_K_2PI)
            ph2 = (float)fmod(pd2, FMD_K_2PI);
          while (ph2 < 0)
            ph2 = (float)((double)ph2 + FMD_K_2PI);
          nco_incr = in2;
          nco_phase = ph2;
          chunk[j & 1][u][lane] = nco_incr;
        };
#ifdef FMD_DBG_NO_FM /* dev aid (tools/ubench/serial_stage): the second wave's loop alone */
        if (true)
        {
          for (unsigned u = 0; u < cnt; u++)
            chunk[j & 1][u][lane] = nco_incr;
        }
        else
#endif
        if (cnt == (unsigned)DS)
        { /* Full chunks: FM_UNROLL samples per trip (no register copies at the back edge, LDS
           * addresses with immediate offsets) and ONE rare-input test per trip: the samples of a group
           * run straight through, their rare flags are collected, and a group in which any lane met a
           * rare input is redone literally from the state it started with (a branch per sample costs
           * three scalar instructions and keeps the scheduler from moving anything across it). */
#pragma unroll 1
          for (unsigned u = 0; u < (unsigned)DS; u += FM_UNROLL)
          {
            const float phase_g = nco_phase, incr_g = nco_incr;
            uint32_t worst = 0; // the group's largest rare measure: one unsigned maximum per sample
#pragma unroll
            for (unsigned v = 0; v < FM_UNROLL; v++)
              worst = max(worst, fm_sample(u + v));
            if (__builtin_expect(FMD_ANY_LANE(worst >= FMD_ATAN_RARE_LIMIT), 0))
            {
              nco_phase = phase_g;
              nco_incr = incr_g;
#pragma unroll 1
              for (unsigned v = 0; v < FM_UNROLL; v++)
                fm_sample_literal(u + v);
            }
          }
        }
        else
        {
#pragma unroll 1
          for (unsigned u = 0; u < cnt; u++)
          {
            const float phase_g = nco_phase, incr_g = nco_incr;
            if (__builtin_expect(FMD_ANY_LANE(fm_sample(u) >= FMD_ATAN_RARE_LIMIT), 0))
            {
              nco_phase = phase_g;
              nco_incr = incr_g;
              fm_sample_literal(u);
            }
          }
        }
      }
      if (PAIRSYNC)
        lds_publish(done_fm, j + 1);
      else
        lds_barrier();
    }
    if (active)
    {
      st.F(F_NCO_PHASE)[c] = nco_phase;
      st.F(F_NCO_INCR)[c] = nco_incr;
    }
  }
  else
  {
    float p_i1 = st.F(F_P_I1)[c], p_i2 = st.F(F_P_I2)[c], p_q1 = st.F(F_P_Q1)[c], p_q2 = st.F(F_P_Q2)[c];
    float p_x1 = st.F(F_P_X1)[c], p_freq = st.F(F_P_FREQ)[c], p_phase = st.F(F_P_PHASE)[c];
    float p_level = 1000.0f; // FmDecode.cpp:147
    float o_re = st.F(F_OSC_RE)[c], o_im = st.F(F_OSC_IM)[c];
    float dc = st.F(F_DC_OFF)[c];
    // the state is in registers before the chunk loop starts: inside it, the only loads in flight
    // are the staged chunk's, and nothing in the sample loop waits for them
    __builtin_amdgcn_s_waitcnt(0);
    FmdSincosP256 p_sc = fmd_sincos_p256_lookup_lds(p_phase, sctab); // pilot NCO: one sample ahead
    float vsum = 0.0f, vsumsq = 0.0f;
    /* The two per-sample stores: wave-uniform row bases plus ONE 32-bit byte offset per lane that
     * advances a row per sample (the row buffers stay below 4 GB).  Padded lanes shadow the last
     * channel -- same state, same input, same results -- so their stores write the very same values to
     * the very same places and need no mask (nor the exec save / branch / restore around it). */
    char* __restrict__ br_rows = reinterpret_cast<char*>(br + (size_t)Hbb * CP); // (baseband, 38 kHz * 2 * baseband)
    char* __restrict__ mix_rows = reinterpret_cast<char*>(mix + (size_t)Hmix * CP);
    unsigned row_off = c * (unsigned)sizeof(float2);
    const unsigned row_step = CP * (unsigned)sizeof(float2);
    // staging (see the chunk loop): this lane's 16 bytes of the rows 4 i + co_row, as 32-bit byte
    // offsets from the chunk's first sample in channel 0's row (the host keeps the buffer below 4 GB)
    constexpr unsigned CO_LPR = DS / 2;      // lanes per row: a chunk of a row is DS * 8 bytes, 16 per lane
    constexpr unsigned CO_RPI = 64 / CO_LPR; // rows per instruction
    const unsigned co_row = lane / CO_LPR, co_col = lane % CO_LPR;
    unsigned co_off[DS / 2];
#pragma unroll
    for (unsigned i = 0; i < DS / 2; i++)
      co_off[i] = min((blockIdx.x * NG + grp) * 64 + CO_RPI * i + co_row, C - 1) *
                      (Mstride * (unsigned)sizeof(float2)) +
                  co_col * 16u;
    for (unsigned j = 0; j <= nchunks; j++)
    {
      /* This wave also moves the FM wave's input: while that wave works on chunk j, the IF-FIR
       * samples of chunk j+1 travel HBM -> registers -> LDS here, a whole chunk ahead of their
       * use.  All DS loads of the chunk are issued before the sample loop and land during it (a
       * load per sample inside the loop would have to return within one iteration, ~0.4 us:
       * no margin against HBM latency once other kernels use the memory system). */
      const unsigned pf0 = (j + 1) * DS; // first sample of the chunk being staged
      const bool staging = (j + 1) < nchunks;
      /* Loads and their wait written by hand.  The compiler cannot
       * count the sample loop's stores, so in front of the LDS writes below it waits for the wave's
       * LAST operations too -- the two stores of the sample just finished, 3-8 us under load --
       * and a second wave that late makes the FM wave wait (seen per workgroup with the probe: up
       * to +13 % cycles, in a third of the workgroups of a launch).  Vector memory operations
       * retire in issue order: with the 2 DS stores of a full chunk behind the DS / 2 loads,
       * `s_waitcnt vmcnt(2 DS - 1)` is enough, and the oldest of those stores is a chunk old.
       * (A ragged last chunk is loaded whole: the host leaves DS samples of slack behind the last
       * channel's row; what lies beyond M is never used.) */
      /* The loads are cooperative: a chunk is 64 rows (channels) of 256 contiguous bytes, and a lane
       * reading its own channel's row 8 bytes at a time touches 64 cache lines per instruction (64
       * cycles of the CU's L1 each: with two groups per CU the L1 was busy a fifth of the time and the
       * stage's stores queued behind it).  Instead 16 lanes read one row 16 bytes each and an instruction
       * covers 4 rows = 8 lines; the transposition happens in the LDS writes below. */
      fmd_v4f pre_c[DS / 2];
      if (staging)
      {
        const float2* sb = demod + pf0; // wave-uniform; the lanes' row offsets are co_off[]
#pragma unroll
        for (unsigned i = 0; i < DS / 2; i += 4)
          asm volatile("global_load_dwordx4 %0, %4, %8\n\t"
                       "global_load_dwordx4 %1, %5, %8\n\t"
                       "global_load_dwordx4 %2, %6, %8\n\t"
                       "global_load_dwordx4 %3, %7, %8"
                       : "=&v"(pre_c[i]), "=&v"(pre_c[i + 1]), "=&v"(pre_c[i + 2]), "=&v"(pre_c[i + 3])
                       : "v"(co_off[i]), "v"(co_off[i + 1]), "v"(co_off[i + 2]), "v"(co_off[i + 3]), "s"(sb)
                       : "memory");
      }
      unsigned stores_behind = 0; // vector stores issued behind those loads
      if (PAIRSYNC && j >= 1) // chunk j - 1 written, stage[(j + 1) & 1] read: FM wave done with j - 1
        lds_wait_ge(done_fm, j, st.spin_limit, st.err);
      if (j >= 1)
      {
        const unsigned m0 = (j - 1) * DS;
        const unsigned cnt = min((unsigned)DS, M - m0);
        float pinc_next = chunk[(j - 1) & 1][0][lane];
        auto second_sample = [&](unsigned u) {
            /* FM PLL output stage (FmDecode.cpp:409-412): low-pass of the NCO frequency term as
             * DC offset, off the PLL's own recurrence and therefore done here.  The chunk entry is
             * read one sample ahead so its LDS latency is not at the head of the iteration. */
            const float pinc = 2 * pinc_next; // phaseIncr (:409), exact
            pinc_next = chunk[(j - 1) & 1][min(u + 1, (unsigned)DS - 1)][lane];
            dc = (float)((1 - 0.0001) * (double)dc + 0.0001 * (double)pinc);
            const float v = (pinc - dc) * k.demod_gain;
            vsum += v;
            vsumsq += v * v;
            /* ---- pilot PLL (FmDecode.cpp:151-217) ---- */
            float ps, pc;
            fmd_sincos_p256_finish(p_sc, m16, &ps, &pc); // looked up when p_phase was formed
            const float tone = 2 * ps * pc;
            float ph_i = ps * v;
            float ph_q = pc * v;
            ph_i = k.p_b0 * ph_i - k.p_a1 * p_i1 - k.p_a2 * p_i2;
            ph_q = k.p_b0 * ph_q - k.p_a1 * p_q1 - k.p_a2 * p_q2;
            p_i2 = p_i1;
            p_i1 = ph_i;
            p_q2 = p_q1;
            p_q1 = ph_q;
            /* :194-201 as selects; the quotient is formed unconditionally and only used in lock */
            const float ratio = ph_q / ph_i;
            const float sgn = (ph_q > 0) ? 1.0f : -1.0f;
            const float perr = (ph_i > fabsf(ph_q)) ? ratio : sgn;
            p_level = (ph_i < p_level) ? ph_i : p_level;
            p_freq += k.p_lf_b0 * perr + k.p_lf_b1 * p_x1;
            p_x1 = perr;
            // :210 std::max(min, std::min(max, freq)): the same as min / max instructions for every
            // input (a NaN frequency becomes maxfreq either way; the limits are positive, no zero signs)
            p_freq = fmaxf(k.p_minfreq, fminf(k.p_maxfreq, p_freq));
            p_phase += p_freq;
            {
              const double pd = (double)p_phase;
              const float down = (float)(pd - FMD_K_2PI);
              p_phase = (pd > FMD_K_2PI) ? down : p_phase; // :215-216
            }
            /* the next sample's table entry: its LDS latency lies under the oscillator and the stores
             * below (the barrier keeps the compiler from moving those in front of the read) */
            p_sc = fmd_sincos_p256_lookup_lds(p_phase, sctab);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MIX)
            {
            /* ---- RDS oscillator mix (DownConvert.cpp:436-442, :464-465), imag(input) == 0 ---- */
            float2 osc;
            osc.x = o_re * k.osc_cos - o_im * k.osc_sin;
            osc.y = o_im * k.osc_cos + o_re * k.osc_sin;
            const float gn = (float)(1.95 - (double)(o_re * o_re + o_im * o_im));
            o_re = gn * osc.x;
            o_im = gn * osc.y;
            const float zero = 0.0f;
#ifdef FMD_DBG_NO_STORES /* dev aid: the loop without its two stores per sample (results kept alive) */
            vsum += tone * (2 * v) + ((v * osc.x) - (zero * osc.y)) + ((v * osc.y) + (zero * osc.x));
#else
            *reinterpret_cast<float2*>(br_rows + row_off) = make_float2(v, tone * (2 * v)); // FmDecode.cpp:456
            *reinterpret_cast<float2*>(mix_rows + row_off) =
                make_float2((v * osc.x) - (zero * osc.y), (v * osc.y) + (zero * osc.x));
#endif
            }
            else
              *reinterpret_cast<float2*>(br_rows + row_off) = make_float2(v, tone * (2 * v)); // FmDecode.cpp:456
            row_off += row_step;
        };
#ifdef FMD_DBG_NO_2ND /* dev aid (tools/ubench/serial_stage): the FM wave's loop alone */
        if (true)
        {
        }
        else
#endif
        if (cnt == (unsigned)DS)
        { // full chunks: two samples per trip (no register copies at the back edge)
#pragma unroll 1
          for (unsigned u = 0; u < (unsigned)DS; u += 2)
          {
            second_sample(u);
            second_sample(u + 1);
          }
        }
        else
        {
#pragma unroll 1
          for (unsigned u = 0; u < cnt; u++)
            second_sample(u);
        }
        stores_behind = (MIX ? 2 : 1) * cnt;
      }
      if (staging)
      {
        if (stores_behind >= (MIX ? 2 : 1) * DS)
        { // all but the chunk's stores, which are younger than the staging loads
          if ((MIX ? 2 : 1) * DS == 64)
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
          else if ((MIX ? 2 : 1) * DS == 32)
            asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (unsigned i = 0; i < DS / 2; i++)
        { // samples 2 * co_col, 2 * co_col + 1 of channel CO_RPI * i + co_row
          stage[(j + 1) & 1][2 * co_col][CO_RPI * i + co_row] = make_float2(pre_c[i].x, pre_c[i].y);
          stage[(j + 1) & 1][2 * co_col + 1][CO_RPI * i + co_row] = make_float2(pre_c[i].z, pre_c[i].w);
        }
      }
      if (PAIRSYNC)
        lds_publish(done_2nd, j + 1);
      else
        lds_barrier();
    }
    if (active)
    {
      st.F(F_P_I1)[c] = p_i1;
      st.F(F_P_I2)[c] = p_i2;
      st.F(F_P_Q1)[c] = p_q1;
      st.F(F_P_Q2)[c] = p_q2;
      st.F(F_P_X1)[c] = p_x1;
      st.F(F_P_FREQ)[c] = p_freq;
      st.F(F_P_PHASE)[c] = p_phase;
      st.F(F_P_LEVEL)[c] = p_level;
      st.F(F_OSC_RE)[c] = MIX ? o_re : osc_after_re;
      st.F(F_OSC_IM)[c] = MIX ? o_im : osc_after_im;
      st.F(F_DC_OFF)[c] = dc;
      { // lock status (FmDecode.cpp:219-228)
        int cnt = st.I(I_P_LOCK_CNT)[c];
        if (2 * p_level > k.p_minsignal)
        {
          if (cnt < k.p_lock_delay)
            cnt += (int)M;
        }
        else
          cnt = 0;
        st.I(I_P_LOCK_CNT)[c] = cnt;
        st.I(I_STEREO)[c] = cnt >= k.p_lock_delay;
        // the audio tail of this call may run after the next call's serial stage: its own copy
        st.I(I_STEREO_Q0 + (int)stereo_q)[c] = cnt >= k.p_lock_delay;
      }
      { // baseband stats (FmDecode.cpp:439-442)
        const float mean = vsum / (float)M;
        const float rms = sqrtf(vsumsq / (float)M);
        st.F(F_BB_MEAN)[c] = 0.95f * st.F(F_BB_MEAN)[c] + 0.05f * mean;
        st.F(F_BB_LEVEL)[c] = 0.95f * st.F(F_BB_LEVEL)[c] + 0.05f * rms;
      }
    }
  }
  if (wg_probe && threadIdx.x == 64) // a pilot/RDS role wave: the last to finish
  {
    wg_probe[3 * blockIdx.x] = probe_r0;
    wg_probe[3 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    // cycles in the low 40 bits; above them where the workgroup ran: HW_ID (bits 8-15: CU, SH, SE)
    // and XCC_ID
    const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    wg_probe[3 * blockIdx.x + 2] = (((long long)__builtin_readcyclecounter() - probe_c0) & 0xffffffffffll) |
                                   ((long long)((hw_id >> 8) & 0xffu) << 40) |
                                   ((long long)(xcc_id & 0xfu) << 48);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K3: CHalfBandDecimateBy2::DecBy2 (DownConvert.cpp:512-550), time-parallel.  in has L-1       */
/*     history rows in front; output k reads rows 2k .. 2k+L-1.  Tap 0 is counted twice and     */
/*     the centre tap added last, like the reference.                                           */
/* ------------------------------------------------------------------------------------------ */
struct HbCoef
{
  float c[52];
  float e[28]; // the even taps c[0], c[2], ... packed (k_halfband4 reads runs of them)
};

#ifndef FMD_HB_R
#define FMD_HB_R 4
#endif
constexpr int HB_R = FMD_HB_R; // outputs per thread: each even input row is loaded once for up to 4 outputs

__global__ __launch_bounds__(256) void k_halfband(const float2* __restrict__ in,
                                                  float2* __restrict__ out, unsigned n_out, int L,
                                                  HbCoef hc, unsigned C, unsigned CP, unsigned Hout)
{
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  // threadIdx.y is the same for all 64 lanes of a wave; saying so keeps tap/table loads scalar
  const unsigned wy = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const unsigned k0 = (blockIdx.y * blockDim.y + wy) * HB_R;
  if (c >= C || k0 >= n_out)
    return;
  const int nr = (int)min((unsigned)HB_R, n_out - k0);
  const int half = (L - 1) / 2; // index of the last even tap is 2*half' with half' = (L-1)/2
  const int mid = half;
  const float2* __restrict__ p = in + (size_t)(2 * k0) * CP + c;
  float ar[HB_R], ai[HB_R];
  // even rows e = 2*k0 + 2*u feed output r with tap j = 2*(u - r), in ascending j per output
  const int nu = half + nr; // u = 0 .. half + nr - 1
  for (int u0 = 0; u0 < nu; u0 += 4)
  {
    float2 xs[4];
#pragma unroll
    for (int q = 0; q < 4; q++) // four independent loads in flight (index clamped, not branched)
      xs[q] = p[(size_t)(2 * min(u0 + q, nu - 1)) * CP];
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
      const int u = u0 + q;
      const float2 x = xs[q];
#pragma unroll
      for (int r = 0; r < HB_R; r++)
      {
        const int jh = u - r;
        if (u < nu && r < nr && jh >= 0 && jh <= half)
        {
          const float cj = hc.c[2 * jh];
          if (jh == 0)
          { // :529-530 tap 0 initialises the accumulator and is then added again in the loop
            ar[r] = x.x * cj;
            ai[r] = x.y * cj;
          }
          ar[r] = ar[r] + x.x * cj;
          ai[r] = ai[r] + x.y * cj;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < HB_R; r++)
  {
    if (r < nr)
    {
      const float2 x = p[(size_t)(2 * r + mid) * CP];
      ar[r] = ar[r] + x.x * hc.c[mid];
      ai[r] = ai[r] + x.y * hc.c[mid];
      out[(size_t)(Hout + k0 + r) * CP + c] = make_float2(ar[r], ai[r]);
    }
  }
}

/* Short blocks.  CHalfBandDecimateBy2::DecBy2 works in place (pInData == pOutData, DownConvert.cpp:
 * 480) and has two regimes below 2 (L - 1) inputs that are part of what the reference computes:
 *  - InLength < L (:519-520): nothing is filtered, the call returns InLength / 2 and the "outputs" are
 *    the first InLength / 2 INPUTS; the delay line is left alone           -> k_hb_pass, no roll
 *  - L <= InLength < 2 (L - 1): filtered as usual, but the delay line is refilled from the in / out
 *    array after the outputs were written over its front (:546-547): entry i is array element
 *    InLength - L + 1 + i, which is an OUTPUT when that index is below the output count
 *                                                                           -> k_roll_hb_mixed */
__global__ void k_hb_pass(const float2* __restrict__ in, unsigned H, float2* __restrict__ out, unsigned Hout,
                          unsigned n_out, unsigned CP)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP)
    return;
  for (unsigned k = blockIdx.y; k < n_out; k += gridDim.y)
    out[(size_t)(Hout + k) * CP + c] = in[(size_t)(H + k) * CP + c];
}

/* dst rows [0, H) <- array elements n - H + r: outputs (rows Hout + idx of `outp`) below n_out, else
 * inputs (rows H + idx of `in`).  dst may be `in` (rows move towards the front: ascending order). */
__global__ void k_roll_hb_mixed(const float2* in, const float2* __restrict__ outp, float2* dst, unsigned H,
                                unsigned n, unsigned n_out, unsigned Hout, unsigned CP)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP)
    return;
  for (unsigned r = 0; r < H; r++)
  {
    const unsigned idx = n - H + r;
    dst[(size_t)r * CP + c] = idx < n_out ? outp[(size_t)(Hout + idx) * CP + c] : in[(size_t)(H + idx) * CP + c];
  }
}

/* CHalfBand11TapDecimateBy2::DecBy2 (DownConvert.cpp:589-688), the first stage when the baseband
 * rate is 320 kHz or more (SetDataRate, :340-341).  Same window indexing as above with L = 11
 * (10 history rows = the class's d0..d9), but a different sum: seven products H0 x0 + H2 x2 + H4 x4 +
 * H5 x5 + H6 x6 + H8 x8 + H10 x10 added left to right as written (:596-661), the centre tap in its
 * place, no tap counted twice; InLength / 2 outputs (an odd last input is only kept as history). */
__global__ __launch_bounds__(256) void k_halfband11(const float2* __restrict__ in,
                                                    float2* __restrict__ out, unsigned n_out, HbCoef hc,
                                                    unsigned C, unsigned CP, unsigned Hout)
{
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  const unsigned wy = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const unsigned o = blockIdx.y * blockDim.y + wy;
  if (c >= C || o >= n_out)
    return;
  const float2* __restrict__ p = in + (size_t)(2 * o) * CP + c;
  const int T[7] = {0, 2, 4, 5, 6, 8, 10};
  float2 x[7];
#pragma unroll
  for (int t = 0; t < 7; t++)
    x[t] = p[(size_t)T[t] * CP];
  float ar = hc.c[0] * x[0].x, ai = hc.c[0] * x[0].y;
#pragma unroll
  for (int t = 1; t < 7; t++)
  {
    ar = ar + hc.c[T[t]] * x[t].x;
    ai = ai + hc.c[T[t]] * x[t].y;
  }
  out[(size_t)(Hout + o) * CP + c] = make_float2(ar, ai);
}

/* ------------------------------------------------------------------------------------------ */
/* K4: cFirFilter::Process(complex) / ProcessTwo (FirFilter.cpp:330-350, :387-413),            */
/*     time-parallel.  The reference walks its ring buffer from slot 0, so output i (global     */
/*     index g = g0 + i since the filter was initialised) sums ages a0, a0+1, ..., T-1, 0, ...  */
/*     with a0 = g mod T, starting from the first product (no leading zero).  in has T-1        */
/*     history rows in front (zeros after init).  I and Q taps are the same table.              */
/* ------------------------------------------------------------------------------------------ */
constexpr int RF_TI = 32; // outputs per workgroup tile

__device__ __forceinline__ float rf_mul(float k, float x) { return k * x; }
__device__ __forceinline__ float2 rf_mul(float k, float2 x) { return make_float2(k * x.x, k * x.y); }
__device__ __forceinline__ void rf_acc(float& a, float k, float x) { a += k * x; }
__device__ __forceinline__ void rf_acc(float2& a, float k, float2 x)
{
  a.x += k * x.x;
  a.y += k * x.y;
}

/* The same filter without per-term tests: thread = (channel lane, RR consecutive outputs).  Output
 * r takes the even rows u = r .. r + half (row u = input row 2*k0 + 2u) with the even taps
 * e[u - r]; so the rows u = RR-1 .. half are taken by every output, with RR taps that are
 * contiguous in e[], and only the first and last RR-1 rows by some.  Tap 0 starts the sum and is
 * added again, the centre tap comes last, like the reference.  Needs half >= RR. */
template <int RR>
__device__ __forceinline__ void hb_group(const float2* __restrict__ in, float2* __restrict__ out,
                                         unsigned k0, int half, const HbCoef& hc, unsigned c, unsigned CP,
                                         unsigned Hout)
{
  const float2* __restrict__ p = in + (size_t)(2 * k0) * CP + c;
  float2 acc[RR];
  const size_t step = (size_t)2 * CP;
#pragma unroll
  for (int u = 0; u < RR; u++) // the rows on which outputs start (u == r: tap 0, twice)
  {
    const float2 x = p[(size_t)u * step];
    acc[u] = rf_mul(hc.e[0], x);
    rf_acc(acc[u], hc.e[0], x);
#pragma unroll
    for (int r = 0; r < u; r++)
      rf_acc(acc[r], hc.e[u - r], x);
  }
#pragma unroll 4
  for (int u = RR; u <= half; u++) // every output: taps e[u], e[u-1], ..., e[u-RR+1]
  {
    const float2 x = p[(size_t)u * step];
#pragma unroll
    for (int r = 0; r < RR; r++)
      rf_acc(acc[r], hc.e[u - r], x);
  }
#pragma unroll
  for (int m = 1; m < RR; m++) // the rows behind the first output's window
  {
    const float2 x = p[(size_t)(half + m) * step];
#pragma unroll
    for (int r = m; r < RR; r++)
      rf_acc(acc[r], hc.e[half + m - r], x);
  }
#pragma unroll
  for (int r = 0; r < RR; r++)
  {
    const float2 x = p[(size_t)(2 * r + half) * CP];
    rf_acc(acc[r], hc.c[half], x);
    out[(size_t)(Hout + k0 + r) * CP + c] = acc[r];
  }
}

__global__ __launch_bounds__(256) void k_halfband4(const float2* __restrict__ in,
                                                   float2* __restrict__ out, unsigned n_out, int L,
                                                   HbCoef hc, unsigned C, unsigned CP, unsigned Hout)
{
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  const unsigned wy = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const unsigned k0 = (blockIdx.y * blockDim.y + wy) * 4;
  if (c >= C || k0 >= n_out)
    return;
  const int half = (L - 1) / 2;
  switch (min(4u, n_out - k0))
  {
    case 4: hb_group<4>(in, out, k0, half, hc, c, CP, Hout); break;
    case 3: hb_group<3>(in, out, k0, half, hc, c, CP, Hout); break;
    case 2: hb_group<2>(in, out, k0, half, hc, c, CP, Hout); break;
    default: hb_group<1>(in, out, k0, half, hc, c, CP, Hout); break;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K3': the three half-band stages of the usual chains as ONE stream (large batches).            */
/*                                                                                              */
/* Three launches of k_halfband4 move the intermediate rows through memory twice (write, read:    */
/* 0.7 GB per call at 8192 channels for 0.44 GB of input and output).  Here a workgroup owns 64  */
/* channels and a stretch of the last stage's outputs and walks it in time order; the outputs of */
/* stage 0 and stage 1 only ever exist in two LDS rings of 64 rows ([row][lane] float2).  A step  */
/* = up to 16 / 8 / 4 outputs of stage 0 / 1 / 2, a group of 4 / 2 / 1 per wave (hb_rows: the     */
/* same sums in the same order as hb_group), two barriers.  The steps of a stretch -- how far      */
/* each stage may run given what its input ring holds and what its output ring can take -- are    */
/* the same for every channel: the host lists them (HbStep).  A stretch that does not start at   */
/* the call's first output computes the 22 + 2 * 42 stage-0 outputs (+ 42 of stage 1) in front of */
/* it again; the call's first rows find the previous call's last outputs in the rings (loaded     */
/* from the history rows of the stage buffers, which the per-stage kernels keep too: the two      */
/* forms can follow each other), and the last outputs of stages 0 and 1 go to `tail1` / `tail2`, */
/* from where the chain's roll moves them into those history rows.  Stage 0's rows are fetched    */
/* a step ahead (15 rows per wave and step in registers).                                         */
/* ------------------------------------------------------------------------------------------ */
/* Behind a call that wrote no mixed rows: the H rows of history the NEXT call's first half-band stage
 * finds in front of its input, should that call take a launch per stage (rows M - H .. M - 1 of
 * baseband x oscillator, as the serial stage's MIX form writes them). */
__global__ void k_mix_tail(const float2* __restrict__ br_last, const float2* __restrict__ osc_last,
                           float2* __restrict__ dst, unsigned H, unsigned CP)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP)
    return;
  for (unsigned r = blockIdx.y; r < H; r += gridDim.y)
  {
    const float v = br_last[(size_t)r * CP + c].x;
    const float2 o = osc_last[r];
    const float zero = 0.0f;
    dst[(size_t)r * CP + c] = make_float2((v * o.x) - (zero * o.y), (v * o.y) + (zero * o.x));
  }
}

struct HbStep
{
  int a_lo, a_n, b_lo, b_n, c_lo, c_n; // outputs of stage 0 / 1 / 2 this step computes (first, count)
  int pad0, pad1;
};
constexpr int HBF_RING = 64; // rows per ring (power of two): >= L - 1 + two steps' outputs of the stage before

/* RR consecutive outputs of one stage from the rows `ld` delivers (row = index into the stage's input
 * with its L - 1 history rows in front: output k takes rows 2k .. 2k + L - 1).  Order of the sum as in
 * hb_group: tap 0 twice, the even taps ascending, the centre tap last (DownConvert.cpp:526-543). */
template <int RR, int HALF, class LD>
__device__ __forceinline__ void hb_rows(LD ld, const HbCoef& hc, float2 (&acc)[RR])
{
  static_assert(HALF >= RR, "half-band group");
#pragma unroll
  for (int u = 0; u < RR + HALF; u++) // even row u: output r takes it with tap e[u - r]
  {
    const float2 x = ld(2 * u);
#pragma unroll
    for (int r = 0; r < RR; r++)
    {
      const int j = u - r;
      if (j == 0)
      {
        acc[r] = rf_mul(hc.e[0], x);
        rf_acc(acc[r], hc.e[0], x);
      }
      else if (j > 0 && j <= HALF)
        rf_acc(acc[r], hc.e[j], x);
    }
  }
#pragma unroll
  for (int r = 0; r < RR; r++)
    rf_acc(acc[r], hc.c[HALF], ld(2 * r + HALF));
}

/* OSC: stage 0's input rows are not the mixed rows but (baseband, -) rows, and row r meets the RDS
 * oscillator's value osc[r] on its way into the sum -- (v osc.x - 0 osc.y, v osc.y + 0 osc.x) like
 * CRDSDownConvert::ProcessData writes it (DownConvert.cpp:464-465; the input's imaginary part is zero). */
template <int H0, int H1, int H2, bool OSC = false>
__global__ __launch_bounds__(256) void k_halfband_chain(
    const float2* __restrict__ mix, const float2* __restrict__ hist1, const float2* __restrict__ hist2,
    float2* __restrict__ out, unsigned Hout, float2* __restrict__ tail1, float2* __restrict__ tail2,
    HbCoef hc0, HbCoef hc1, HbCoef hc2, const HbStep* __restrict__ steps, const int* __restrict__ seg_first,
    unsigned n_in, unsigned n0, unsigned n1, unsigned C, unsigned CP, const float2* __restrict__ osc,
    unsigned prio)
{
  wave_prio(prio);
  __shared__ float2 ring1[HBF_RING][64]; // stage 0's outputs, row i0 (>= -2 H1: history) at slot i0 & 63
  __shared__ float2 ring2[HBF_RING][64]; // stage 1's outputs
  constexpr int L1H = 2 * H1, L2H = 2 * H2; // history rows of stages 1 and 2
  static_assert(L1H + 34 <= HBF_RING && L2H + 18 <= HBF_RING, "ring size");
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const unsigned c0 = blockIdx.x * 64 + lane;
  const bool live = c0 < C;
  const unsigned c = live ? c0 : C - 1;
  const int s_begin = seg_first[blockIdx.y], s_end = seg_first[blockIdx.y + 1];
  if (s_begin >= s_end)
    return;
  const float2* __restrict__ mp = mix + c;
  const size_t rowstride = CP;
  // stage 0's input rows of this wave's group of a step: even rows 0, 2, .. 2 (3 + H0) and the four centres
  constexpr int NA = 4 + H0 + 4;
  constexpr int NSET = 4; // register sets: the rows of a step are fetched NSET - 1 steps ahead
  float2 xs[NSET][NA];
  /* Always all fifteen loads, rows clamped, never branched (a group at the end of the input has fewer
   * than four outputs, a step may have none for this wave): the compiler can only wait for "all but the
   * N youngest" loads, and it knows N -- the three younger sets that are still in flight -- only if every
   * path issues the same number. */
  auto fetch_a = [&](float2 (&x)[NA], const HbStep& st) {
    const int k0 = st.a_lo + 4 * w;
    const int last = 2 * H0 + (int)n_in - 1;
#pragma unroll
    for (int u = 0; u < 4 + H0; u++)
      x[u] = mp[(size_t)min(2 * k0 + 2 * u, last) * rowstride];
#pragma unroll
    for (int r = 0; r < 4; r++)
      x[4 + H0 + r] = mp[(size_t)min(2 * k0 + 2 * r + H0, last) * rowstride];
  };
  // the rings' history (the first stretch of a call): rows -L1H .. -1 / -L2H .. -1
  {
    const HbStep f = steps[s_begin];
    if (2 * f.b_lo - L1H < 0)
      for (int i = w; i < L1H; i += 4)
        ring1[(i - L1H) & (HBF_RING - 1)][lane] = hist1[(size_t)i * rowstride + c];
    if (2 * f.c_lo - L2H < 0)
      for (int i = w; i < L2H; i += 4)
        ring2[(i - L2H) & (HBF_RING - 1)][lane] = hist2[(size_t)i * rowstride + c];
#pragma unroll
    for (int k = 0; k < NSET - 1; k++) // (a stretch's list has a multiple of NSET steps, empty ones at its end)
      fetch_a(xs[k], steps[s_begin + k]);
  }
  __syncthreads();
  // the step lists travel a step ahead of their use too (a scalar load is a round trip to the L2 for a lone wave)
  HbStep cur = steps[s_begin], far = steps[min(s_begin + NSET - 1, s_end - 1)];
  auto step = [&](int s, float2 (&x)[NA], float2 (&xn)[NA]) { // step s out of x; step s + NSET - 1's rows into xn
    const HbStep st = cur;
    const HbStep cur_next = steps[min(s + 1, s_end - 1)], far_next = steps[min(s + NSET, s_end - 1)];
    fetch_a(xn, far);
    { // stage 0: four outputs per wave out of registers
      const int k0 = st.a_lo + 4 * w;
      const int nr = min(4, st.a_lo + st.a_n - k0);
      if (nr > 0)
      {
        float2 acc[4];
        hb_rows<4, H0>(
            [&](int row) {
              const float2 v = (row & 1) ? x[4 + H0 + (row - H0) / 2] : x[row / 2];
              if constexpr (!OSC)
                return v;
              else
              {
                const float2 o = osc[min(2 * k0 + row, 2 * H0 + (int)n_in - 1)]; // wave-uniform: a scalar load
                const float zero = 0.0f;
                return make_float2((v.x * o.x) - (zero * o.y), (v.x * o.y) + (zero * o.x));
              }
            },
            hc0, acc);
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (r < nr)
          {
            const int i0 = k0 + r;
            ring1[i0 & (HBF_RING - 1)][lane] = acc[r];
            if (live && i0 >= (int)n0 - L1H)
              tail1[(size_t)(i0 - ((int)n0 - L1H)) * rowstride + c] = acc[r];
          }
      }
    }
    lds_barrier();
    { // stage 1: two outputs per wave out of ring 1 (row = output index of stage 0 + L1H)
      const int k0 = st.b_lo + 2 * w;
      const int nr = min(2, st.b_lo + st.b_n - k0);
      if (nr > 0)
      {
        float2 acc[2];
        hb_rows<2, H1>([&](int row) { return ring1[(2 * k0 + row - L1H) & (HBF_RING - 1)][lane]; }, hc1, acc);
#pragma unroll
        for (int r = 0; r < 2; r++)
          if (r < nr)
          {
            const int i1 = k0 + r;
            ring2[i1 & (HBF_RING - 1)][lane] = acc[r];
            if (live && i1 >= (int)n1 - L2H)
              tail2[(size_t)(i1 - ((int)n1 - L2H)) * rowstride + c] = acc[r];
          }
      }
    }
    lds_barrier();
    { // stage 2: one output per wave out of ring 2
      const int k0 = st.c_lo + w;
      if (k0 < st.c_lo + st.c_n)
      {
        float2 acc[1];
        hb_rows<1, H2>([&](int row) { return ring2[(2 * k0 + row - L2H) & (HBF_RING - 1)][lane]; }, hc2, acc);
        if (live)
          out[(size_t)(Hout + (unsigned)k0) * rowstride + c] = acc[0];
      }
    }
    cur = cur_next;
    far = far_next;
  };
  for (int s = s_begin; s < s_end; s += NSET)
  {
#pragma unroll
    for (int k = 0; k < NSET; k++)
      step(s + k, xs[k], xs[(k + NSET - 1) % NSET]);
  }
}

/* Workgroup = 64 channels x RF_TI outputs.  The T-1+RF_TI input rows of the tile are staged once
 * in LDS ([row][channel]: conflict-free reads), because every input row is needed by T different
 * outputs and re-reading it from L2 for each made the kernel L2-bandwidth bound.
 * E = float2 for the complex / two-stream filters, float for the RDS matched filter. */
template <typename E>
__global__ __launch_bounds__(256) void k_ring_fir(const E* __restrict__ in, E* __restrict__ out,
                                                  unsigned n, int T, const float* __restrict__ taps,
                                                  unsigned g0, unsigned C, unsigned CP, unsigned Hout)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char rtile_raw[];
  E* rtile = reinterpret_cast<E*>(rtile_raw); // [T - 1 + RF_TI][64]
  const unsigned lane = threadIdx.x;
  const unsigned y = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y); // 0..3, wave-uniform
  const unsigned c0 = blockIdx.x * 64 + lane;
  const unsigned c = c0 < C ? c0 : C - 1;
  const unsigned i0 = blockIdx.y * RF_TI;
  const unsigned nt = min((unsigned)RF_TI, n - i0);
  const unsigned rows = (unsigned)T - 1 + nt;
  // buffer row of x[i - a] is (T-1 + i - a); the tile starts at buffer row i0
  for (unsigned r = y; r < rows; r += 4)
    rtile[r * 64 + lane] = in[(size_t)(i0 + r) * CP + c];
  __syncthreads();
  if (c0 >= C)
    return;
  for (unsigned q = y; q < nt; q += 4)
  {
    const unsigned i = i0 + q;
    const int a0 = (int)((g0 + i) % (unsigned)T);
    // newest sample (age 0) sits at tile row T-1+q; age a at row T-1+q-a
    const E* base = rtile + (size_t)((unsigned)T - 1 + q) * 64 + lane;
    E acc = rf_mul(taps[a0], base[-(ptrdiff_t)a0 * 64]);
#pragma unroll 4
    for (int a = a0 + 1; a < T; a++) // ages a0+1 .. T-1
      rf_acc(acc, taps[a], base[-(ptrdiff_t)a * 64]);
#pragma unroll 4
    for (int a = 0; a < a0; a++) // then the ring wraps: ages 0 .. a0-1
      rf_acc(acc, taps[a], base[-(ptrdiff_t)a * 64]);
    out[(size_t)(Hout + i) * CP + c] = acc;
  }
}

/* The same filter for the two float2 instances on the heavy part of the post chain (RDS low-pass,
 * audio low-pass), without LDS and without a barrier: thread = (channel lane, RG consecutive
 * outputs), rows straight from L2 / L1.  Output i sums times B, B-1, ..., i-T+1 and then i, i-1,
 * ..., B+1 with B = i - ((g0 + i) mod T), the time of the sample in ring slot 0; consecutive
 * outputs of one ring period share B, so a group walks the rows all of its outputs take once
 * (four taps per row, contiguous in the table: age = output - row), and the few rows only some of
 * them take on their own.  Every output's accumulator starts at -0 (x + -0 = x for every x), the
 * order is the reference's.  A group that straddles a ring period is done as two groups. */
#ifndef FMD_RG
#define FMD_RG 4
#endif
constexpr int RG = FMD_RG;

__device__ __forceinline__ float rf_neg_zero(float*) { return -0.0f; }
__device__ __forceinline__ float2 rf_neg_zero(float2*) { return make_float2(-0.0f, -0.0f); }

template <int RR, typename E>
__device__ __forceinline__ void ring_group(const E* __restrict__ in, E* __restrict__ out,
                                           unsigned i, int T, const float* __restrict__ taps,
                                           unsigned g0, unsigned c, unsigned CP, unsigned Hout, bool store)
{
  const int a0 = (int)((g0 + i) % (unsigned)T); // a0 + RR - 1 <= T - 1: one ring period
  E acc[RR];
#pragma unroll
  for (int r = 0; r < RR; r++)
    acc[r] = rf_neg_zero((E*)nullptr);
  // buffer row of time t is T - 1 + t
  const E* __restrict__ p1 = in + (size_t)((unsigned)T - 1 + i - (unsigned)a0) * CP + c; // time B
  const float* __restrict__ k1 = taps + a0;
  const int n1 = T - a0 - (RR - 1); // rows B .. i+RR-T, taken by every output: ages a0 + r + s
#pragma unroll 8
  for (int s = 0; s < n1; s++)
  {
    const E x = *p1;
    p1 -= CP;
#pragma unroll
    for (int r = 0; r < RR; r++)
      rf_acc(acc[r], k1[s + r], x);
  }
#pragma unroll
  for (int m = 0; m < RR - 1; m++) // the oldest rows: output r takes RR-1-r of them, up to age T-1
  {
    const E x = *p1;
    p1 -= CP;
#pragma unroll
    for (int r = 0; r < RR - 1 - m; r++)
      rf_acc(acc[r], taps[T - (RR - 1) + r + m], x);
  }
  const E* __restrict__ p2 = in + (size_t)((unsigned)T - 1 + i + RR - 1) * CP + c; // time i+RR-1
#pragma unroll
  for (int m = 0; m < RR - 1; m++) // the newest rows: output r takes the last r of them, from age 0
  {
    const E x = *p2;
    p2 -= CP;
#pragma unroll
    for (int r = RR - 1 - m; r < RR; r++)
      rf_acc(acc[r], taps[r - (RR - 1 - m)], x);
  }
#pragma unroll 8
  for (int s = 0; s < a0; s++) // rows i .. B+1, taken by every output: ages r + s
  {
    const E x = *p2;
    p2 -= CP;
#pragma unroll
    for (int r = 0; r < RR; r++)
      rf_acc(acc[r], taps[s + r], x);
  }
  if (store)
  {
#pragma unroll
    for (int r = 0; r < RR; r++)
      out[(size_t)(Hout + i + r) * CP + c] = acc[r];
  }
}

template <int RR, typename E>
__device__ __forceinline__ void ring_dispatch(unsigned take, const E* __restrict__ in,
                                              E* __restrict__ out, unsigned i, int T,
                                              const float* __restrict__ taps, unsigned g0, unsigned c,
                                              unsigned CP, unsigned Hout, bool store)
{ // take is wave-uniform: one scalar branch per size
  if (take == (unsigned)RR)
    ring_group<RR, E>(in, out, i, T, taps, g0, c, CP, Hout, store);
  else if constexpr (RR > 1)
    ring_dispatch<RR - 1, E>(take, in, out, i, T, taps, g0, c, CP, Hout, store);
}

template <typename E>
__global__ __launch_bounds__(256) void k_ring_fir4(const E* __restrict__ in, E* __restrict__ out,
                                                   unsigned n, int T, const float* __restrict__ taps,
                                                   unsigned g0, unsigned C, unsigned CP, unsigned Hout,
                                                   unsigned prio)
{
  // the real instance is the matched filter between two lane-per-channel kernels of the light part:
  // short, and the light part should be over before the next FIR starts -> issue first, like them (prio 3)
  wave_prio(prio);
  const unsigned c = blockIdx.x * 64 + threadIdx.x; // < CP: the row buffers are padded
  const unsigned y = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  unsigned i = (blockIdx.y * blockDim.y + y) * RG;
  if (i >= n)
    return;
  const bool store = c < C;
  unsigned left = min((unsigned)RG, n - i);
  while (left)
  { // as many outputs as stay within one ring period
    const unsigned room = (unsigned)T - (g0 + i) % (unsigned)T;
    const unsigned take = min(left, room);
    ring_dispatch<RG, E>(take, in, out, i, T, taps, g0, c, CP, Hout, store);
    i += take;
    left -= take;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K5: RDS recurrences at the RDS rate.  The matched filter between the two serial kernels     */
/*     (cFirFilter::Process(real), FirFilter.cpp:360-377) runs time-parallel in k_ring_fir.      */
/* ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ uint32_t rds_check_block(uint32_t& in_bits, uint32_t offset, bool fec)
{
  const uint32_t parckh[16] = {0x2DC, 0x16E, 0x0B7, 0x287, 0x39F, 0x313, 0x355, 0x376,
                               0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B};
  uint32_t tb = 0x3FFFFFF & in_bits;
  uint32_t syn = tb >> 16;
#pragma unroll
  for (int i = 0; i < 16; i++)
  {
    if (tb & 0x8000)
      syn ^= parckh[i];
    tb <<= 1;
  }
  syn ^= offset;
  if (syn && fec)
  {
    uint32_t mask = 1u << 25;
    for (int i = 0; i < 16; i++)
    {
      if (syn & 0x200)
      {
        if ((syn & 0x1F) == 0)
        {
          in_bits ^= mask;
          syn <<= 1;
        }
        else
        {
          syn <<= 1;
          syn ^= 0x5B9;
        }
      }
      else
        syn <<= 1;
      mask >>= 1;
    }
    syn &= 0x3FF;
  }
  return syn;
}

/* K5a: ProcessRdsPll (RDSProcess.cpp:222-270), one lane per channel.  Output = de-rotated
 *      imaginary part, written behind the T_mf-1 history rows the matched filter needs.
 *      Four waves (one per SIMD) share the 16 KB sine / cosine table of a workgroup: a quarter as many
 *      CUs carry one during the 0.2-0.5 ms the kernel runs, which matters to the whole-CU resampler. */
constexpr int RP_WAVES = 4;
__global__ __launch_bounds__(64 * RP_WAVES) void k_rds_pll(const float2* __restrict__ lpf, unsigned R, unsigned C,
                                                unsigned CP, RdsConsts k, ChannelState st,
                                                float* __restrict__ rpll, unsigned Hout,
                                                const double* __restrict__ sctab_g, FmdSincosTab sct)
{
  __shared__ double sctab[2 * FMD_SINCOS_TAB_SIZE];
  __builtin_amdgcn_s_setprio(3);
  for (unsigned i = threadIdx.y * 64 + threadIdx.x; i < 2 * FMD_SINCOS_TAB_SIZE; i += 64 * RP_WAVES)
    sctab[i] = sctab_g[i];
  __syncthreads();
  const unsigned c = (blockIdx.x * RP_WAVES + threadIdx.y) * 64 + threadIdx.x;
  if (c >= C)
    return;
  float phase = st.F(F_R_PHASE)[c], freq = st.F(F_R_FREQ)[c];
  float* __restrict__ o = rpll + (size_t)Hout * CP + c;
  /* The input travels a whole tile ahead of its use: the loads of tile n + 1 are in flight while
   * tile n goes through the recurrence (one load per sample, issued one sample ahead, had to come
   * back within an iteration -- 0.2 us; beside the bandwidth kernels a load takes several times that
   * and the kernel took 0.56 ms inside the pipeline against 0.23 ms alone). */
  constexpr unsigned PT = 16;
  float2 nxt[PT];
#pragma unroll
  for (unsigned u = 0; u < PT; u++)
    nxt[u] = lpf[(size_t)min(u, R - 1) * CP + c];
  for (unsigned i0 = 0; i0 < R; i0 += PT)
  {
    float2 cur[PT];
#pragma unroll
    for (unsigned u = 0; u < PT; u++)
      cur[u] = nxt[u];
#pragma unroll
    for (unsigned u = 0; u < PT; u++) // clamped: past the end the last row again (never used)
      nxt[u] = lpf[(size_t)min(i0 + PT + u, R - 1) * CP + c];
    const unsigned cnt = min(PT, R - i0);
#pragma unroll
    for (unsigned u = 0; u < PT; u++)
    {
      if (u < cnt)
      {
        const float2 in = cur[u];
        float sn, cs;
        fmd_sincos_tab(phase, sctab, sct, &sn, &cs);
        const float tr = cs * in.x - sn * in.y;
        const float ti = cs * in.y + sn * in.x;
        const float err = -fmd_rds_arctan2(ti, tr);
        freq += (k.pll_beta * err);
        freq = (freq > k.nco_hl) ? k.nco_hl : ((freq < k.nco_ll) ? k.nco_ll : freq);
        phase += (freq + k.pll_alpha * err);
        *o = ti;
        o += CP;
      }
    }
  }
  st.F(F_R_PHASE)[c] = fmodf(phase, (float)FMD_K_2PI); // RDSProcess.cpp:269
  st.F(F_R_FREQ)[c] = freq;
}

/* K5b: after the matched filter (k_ring_fir<float>): squaring + bit-sync resonator
 *      (RDSProcess.cpp:137-142, IirFilter.cpp:78-87), peak slicer (:144-179), ProcessNewRdsBit
 *      (:272-375) and CheckBlock with Meggitt FEC (:377-431).  One lane per channel.  Sliced
 *      bits are queued per lane and the block-sync state machine drains the queue once per
 *      RB_TILE samples, so the wave does not run it on every sample just because some lane has
 *      a bit. */
constexpr int RB_TILE = 32;

__global__ __launch_bounds__(64) void k_rds_bits(const float* __restrict__ mf, unsigned R, unsigned C,
                                                 unsigned CP, RdsConsts k, ChannelState st,
                                                 uint32_t call_index, RdsGroupRec* __restrict__ queue,
                                                 unsigned* __restrict__ queue_count, unsigned queue_cap,
                                                 float* __restrict__ tap_sync, int write_taps)
{
  __builtin_amdgcn_s_setprio(3);
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C)
    return;
  const uint32_t offs[8] = {0x3D8, 0x3D4, 0x25C, 0x258, 0x3D8, 0x3D4, 0x3CC, 0x258};
  float w1 = st.F(F_R_W1)[c], w2 = st.F(F_R_W2)[c];
  float last_sync = st.F(F_R_LAST_SYNC)[c], last_slope = st.F(F_R_LAST_SLOPE)[c],
        last_data = st.F(F_R_LAST_DATA)[c];
  int last_bit = st.I(I_R_LAST_BIT)[c];
  uint32_t bits = (uint32_t)st.I(I_R_BITS)[c];
  int block = st.I(I_R_BLOCK)[c], bitpos = st.I(I_R_BITPOS)[c], state = st.I(I_R_STATE)[c],
      boff = st.I(I_R_BOFF)[c], errors = st.I(I_R_ERRORS)[c];
  uint16_t bd[4];
#pragma unroll
  for (int q = 0; q < 4; q++)
    bd[q] = st.r_data[(size_t)q * CP + c];
  uint32_t seq = (uint32_t)st.I(I_R_SEQ)[c];

  float dnext[RB_TILE];
#pragma unroll
  for (unsigned u = 0; u < RB_TILE; u++)
    dnext[u] = mf[(size_t)min(u, R - 1) * CP + c];
  for (unsigned i0 = 0; i0 < R; i0 += RB_TILE)
  {
    const unsigned cnt = min((unsigned)RB_TILE, R - i0);
    float din[RB_TILE];
#pragma unroll
    for (unsigned u = 0; u < RB_TILE; u++)
      din[u] = dnext[u];
#pragma unroll
    for (unsigned u = 0; u < RB_TILE; u++) // the next tile's loads are in flight during this tile's recurrence
      dnext[u] = mf[(size_t)min(i0 + RB_TILE + u, R - 1) * CP + c];
    uint64_t qbits = 0; // bits sliced in this tile, oldest in the MSBs
    int qcount = 0;
#pragma unroll
    for (unsigned u = 0; u < RB_TILE; u++)
    {
      if (u >= cnt)
        break;
      const float d = din[u];
      const float mag = d * d;
      const float w0 = mag - k.bs_a1 * w1 - k.bs_a2 * w2;
      const float sv = k.bs_b0 * w0 + k.bs_b1 * w1 + k.bs_b2 * w2;
      w2 = w1;
      w1 = w0;
      if (write_taps)
        tap_sync[(size_t)(i0 + u) * CP + c] = sv;
      const float slope = sv - last_sync;
      last_sync = sv;
      if ((slope < 0.0f) && (last_slope * slope) < 0.0f)
      { // top of the sync sine: read the previous matched-filter sample, differential decode
        const int bit = (last_data >= 0) ? 1 : 0;
        qbits = (qbits << 1) | (uint64_t)(bit ^ last_bit);
        qcount++;
        last_bit = bit;
      }
      last_data = d;
      last_slope = slope;
    }

    while (__any(qcount > 0))
    {
      if (qcount > 0)
      {
        qcount--;
        const uint32_t nb = (uint32_t)((qbits >> qcount) & 1u);
        bits = (bits << 1) | nb;
        bool emit = false;
        if (state == 0)
        { // BITSYNC: look for a clean block A at every bit position
          if (!rds_check_block(bits, offs[0], false))
          {
            bitpos = 0;
            boff = 0;
            bd[0] = (uint16_t)(bits >> 10);
            block = 1;
            state = 1;
          }
        }
        else if (++bitpos >= 26)
        {
          bitpos = 0;
          if (state == 3)
          { // GROUPRESYNC: skip to the start of the next group
            if (++block > 3)
            {
              block = 0;
              state = 2;
            }
          }
          else
          {
            const uint32_t bad = rds_check_block(bits, offs[block + boff], state == 2);
            if (bad)
            {
              if (state == 1)
                state = 0;
              else
              {
                errors++;
                if (errors > 0) // BLOCK_ERROR_LIMIT 0
                  state = 0;
                else
                {
                  if (++block > 3)
                    block = 0;
                  if (block != 0)
                    state = 3;
                }
              }
            }
            else
            {
              const uint16_t word = (uint16_t)(bits >> 10);
              if (block == 0)
                bd[0] = word;
              else if (block == 1)
                bd[1] = word;
              else if (block == 2)
                bd[2] = word;
              else
                bd[3] = word;
              boff = (block == 1 && (word & 0x0800)) ? 4 : 0;
              if (state == 1)
              { // BLOCKSYNC: four good blocks in sequence confirm the bit position
                if (block >= 3)
                {
                  block = 0;
                  errors = 0;
                  state = 2;
                  emit = true;
                }
                else
                  block++;
              }
              else if (++block > 3)
              { // GROUPDECODE: a complete group
                block = 0;
                errors = 0;
                emit = true;
              }
            }
          }
        }
        if (emit)
        {
          const unsigned slot = atomicAdd(queue_count, 1u);
          if (slot < queue_cap)
          {
            RdsGroupRec r;
            r.channel = c;
            r.call_index = call_index;
            r.seq = seq;
            r.blocks[0] = bd[0];
            r.blocks[1] = bd[1];
            r.blocks[2] = bd[2];
            r.blocks[3] = bd[3];
            queue[slot] = r;
          }
          else
            dev_error(st.err + 1, DEVERR_RDS_QUEUE_FULL);
          seq++;
        }
      }
    }
  }

  st.F(F_R_W1)[c] = w1;
  st.F(F_R_W2)[c] = w2;
  st.F(F_R_LAST_SYNC)[c] = last_sync;
  st.F(F_R_LAST_SLOPE)[c] = last_slope;
  st.F(F_R_LAST_DATA)[c] = last_data;
  st.I(I_R_LAST_BIT)[c] = last_bit;
  st.I(I_R_BITS)[c] = (int)bits;
  st.I(I_R_BLOCK)[c] = block;
  st.I(I_R_BITPOS)[c] = bitpos;
  st.I(I_R_STATE)[c] = state;
  // the status snapshot's RDS state (not a cFmDecoder getter) is this kernel's to write: a word of its
  // own, outside the audio tail's sequence-locked record, so that the tail need not wait for the RDS
  // chain where the two run on different streams
  st.ds[(size_t)HS_R_STATE * st.CP + c] = (unsigned)state;
  st.I(I_R_BOFF)[c] = boff;
  st.I(I_R_ERRORS)[c] = errors;
#pragma unroll
  for (int q = 0; q < 4; q++)
    st.r_data[(size_t)q * CP + c] = bd[q];
  st.I(I_R_SEQ)[c] = (int)seq;
}

/* ------------------------------------------------------------------------------------------ */
/* K6/K7: cDownsampleFilter::Process(real), fractional branch (DownConvert.cpp:195-233).       */
/*     The interpolated tap k_j = coeff[j]*k0 + coeff[j+1]*k1 depends only on the output index  */
/*     (positions are batch-uniform), so it is tabulated once per call (k_rs_table) and the     */
/*     filter proper is a plain per-output dot product over the window, in j order.             */
/* ------------------------------------------------------------------------------------------ */
__global__ void k_rs_table(const float* __restrict__ coeff, unsigned order, float p, float pstep,
                           unsigned A, float* __restrict__ ktab, unsigned row_stride, unsigned margin,
                           int* __restrict__ pidx)
{ // row i = [margin zeros][k_0 .. k_order][margin zeros]; the margins are never written (zero since
  // allocation): a tap index outside 0..order reads an exact zero
  const unsigned i = blockIdx.x;
  if (i >= A)
    return;
  const float pf = p + (float)i * pstep;
  const int pi = (int)pf;
  const float k1 = pf - (float)pi;
  const float k0 = 1 - k1;
  for (unsigned j = threadIdx.x; j <= order; j += blockDim.x)
    ktab[(size_t)i * row_stride + margin + j] = coeff[j] * k0 + coeff[j + 1] * k1;
  if (threadIdx.x == 0)
    pidx[i] = pi;
}

#ifndef FMD_RS_R
#define FMD_RS_R 6 // inside the pipeline 6-7 outputs per thread beat 4, 5 and 8 (+2.7 % whole path; alone all ~0.42 ms)
#endif
#ifndef FMD_RS_B
#define FMD_RS_B 8
#endif
constexpr int RS_R = FMD_RS_R; // outputs per thread; consecutive windows are ~4.5 rows apart, 219 rows long
constexpr int RS_B = FMD_RS_B; // rows per batch: RS_R * RS_B taps live in SGPRs at a time
/* Zero entries the host leaves before and after every output's taps in the table: a wave reads the
 * taps of a whole batch for all of its outputs, up to RS_B - 1 + (RS_R - 1) * ceil(step) entries
 * outside an output's 0..order. */
inline unsigned rs_table_margin(float step)
{
  return unsigned(RS_B + (RS_R - 1) * (int(step) + 2) + 7) / 8 * 8;
}

/* Thread = (channel lane, group of RS_R consecutive outputs).  The union of the group's windows is
 * walked once from the newest row down in batches of RS_B rows; row `top - t` feeds output r with
 * tap j = t - off_r (off_r = top - pidx[r]), so every output still accumulates in ascending j like
 * the reference.  Rows of the union outside an output's own window meet a ZERO tap from the table's
 * margins: the product is +-0 and leaves the sum as it is, bit for bit (a sum that started at +0
 * is never -0), so every batch runs the same test-free code with wide scalar tap loads.  That holds
 * for finite samples only; a batch at the edge of the windows that holds an infinity or a NaN
 * (0 * inf = NaN) takes the literal per-output tests instead (wave-uniform, never in practice).
 * in = (baseband, raw-stereo) pairs, so both resamplers share each load and each tap. */
__global__ __launch_bounds__(256) void k_resample(const float2* __restrict__ br, unsigned Hbb,
                                                  unsigned order, const float* __restrict__ ktab,
                                                  unsigned row_stride, unsigned margin,
                                                  const int* __restrict__ pidx, unsigned A,
                                                  float2* __restrict__ out, unsigned Hout, unsigned C,
                                                  unsigned CP)
{
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  const unsigned wy = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y); // wave-uniform
  const unsigned i0 = (blockIdx.y * blockDim.y + wy) * RS_R;
  if (i0 >= A)
    return;
  const int nr = (int)min((unsigned)RS_R, A - i0);
  int off[RS_R];
  const float* kp[RS_R]; // kp[r][t] = tap of output r for window row t
  const int top = pidx[i0 + nr - 1];
#pragma unroll
  for (int r = 0; r < RS_R; r++)
  { // a partial last group computes its last output more than once (not stored)
    const int rr = r < nr ? r : nr - 1;
    off[r] = top - pidx[i0 + rr];
    kp[r] = ktab + (size_t)(i0 + rr) * row_stride + margin - off[r];
  }
  float2 acc[RS_R];
#pragma unroll
  for (int r = 0; r < RS_R; r++)
    acc[r] = make_float2(0.0f, 0.0f);
  // wave-uniform row pointer + 32-bit lane offset (c < CP: the row buffers are padded to CP lanes,
  // and the host keeps RS_B rows of zeros in front of row 0 for the last batch's overhang)
  const char* __restrict__ rp = reinterpret_cast<const char*>(br + (size_t)(Hbb + (unsigned)top) * CP);
  const unsigned lane_off = c * (unsigned)sizeof(float2);
  const size_t row_bytes = (size_t)CP * sizeof(float2);
  const int off0 = off[0];            // largest offset (oldest output of the group)
  const int tend = off0 + (int)order; // last row of the union window

  // the rows of the next batch are fetched while this one is accumulated
  float2 xn[RS_B];
#pragma unroll
  for (int q = 0; q < RS_B; q++)
  {
    xn[q] = *reinterpret_cast<const float2*>(rp + lane_off);
    rp -= row_bytes;
  }
  for (int t = 0; t <= tend; t += RS_B)
  {
    float2 xs[RS_B];
#pragma unroll
    for (int q = 0; q < RS_B; q++)
      xs[q] = xn[q];
#pragma unroll
    for (int q = 0; q < RS_B; q++) // past the end of the window: rows nobody takes (zero taps)
    {
      xn[q] = *reinterpret_cast<const float2*>(rp + lane_off);
      rp -= row_bytes;
    }
    float kk[RS_R][RS_B];
#pragma unroll
    for (int r = 0; r < RS_R; r++)
    {
#pragma unroll
      for (int q = 0; q < RS_B; q++)
        kk[r][q] = kp[r][t + q];
    }
    bool literal = false;
    if (!(t >= off0 && t + RS_B - 1 <= (int)order))
    { // a batch with rows outside some output's window: zero taps are only exact for finite samples
      bool fin = true;
#pragma unroll
      for (int q = 0; q < RS_B; q++)
        fin = fin && __builtin_isfinite(xs[q].x) && __builtin_isfinite(xs[q].y);
      literal = FMD_ANY_LANE(!fin);
    }
    if (!literal)
    {
#pragma unroll
      for (int q = 0; q < RS_B; q++)
      {
#pragma unroll
        for (int r = 0; r < RS_R; r++)
        {
          acc[r].x += kk[r][q] * xs[q].x;
          acc[r].y += kk[r][q] * xs[q].y;
        }
      }
    }
    else
    {
#pragma unroll
      for (int q = 0; q < RS_B; q++)
      {
#pragma unroll
        for (int r = 0; r < RS_R; r++)
        {
          const int j = t + q - off[r];
          if (j >= 0 && j <= (int)order)
          {
            acc[r].x += kk[r][q] * xs[q].x;
            acc[r].y += kk[r][q] * xs[q].y;
          }
        }
      }
    }
  }
  if (c < C)
  {
#pragma unroll
    for (int r = 0; r < RS_R; r++)
      if (r < nr) // (stereo, mono) = ProcessTwo's (A, B): x came from baseband -> mono
        out[(size_t)(Hout + i0 + r) * CP + c] = make_float2(acc[r].y, acc[r].x);
  }
}

/* Measured and dropped: the rows of a workgroup's 16 outputs staged once through LDS (double-buffered
 * batches, one barrier each) instead of every wave fetching its own window from L2: 3.2 x fewer L2
 * reads, 0.42 instead of 0.46 ms alone, but no faster inside the pipeline at 8192 channels and 7 %
 * slower at 32768 (waves idle at the barriers outside their own window). */

/* ------------------------------------------------------------------------------------------ */
/* K6'/K7': the same two resamplers as ONE STREAM over an LDS ring (large batches).             */
/*                                                                                              */
/* k_resample above lets every wave fetch its own 240-row window through L1 / L2: at 8192       */
/* channels the ~1100 workgroups in flight span 76 MB of rows, the L2s hold 32 MB, and every    */
/* row crosses the fabric 5.8 times.  Here a workgroup owns 64 channels and a third (1 / S) of   */
/* the call's outputs and walks them in time order: the rows it needs live in a ring in LDS      */
/* (NBR batches of 8 rows x 512 B, up to 160 KB -- the whole CU), every row is fetched from      */
/* memory ONCE per segment (1 + 219 / (4.55 * outputs per segment) = 1.11 at S = 3).  A step =   */
/* NW * R outputs: wave w adds up outputs R (s NW + w) .. + R - 1 over the union of their        */
/* windows (rs_walk_asm: taps scalar, R outputs share every row read), the rows of the next      */
/* step are fetched into registers meanwhile and go into the ring between two barriers.          */
/* Absolute row rr = call row + RB (RB a multiple of 8 >= the history rows + 8, so that batch    */
/* borders do not move with the call); batch = rr / 8 lives in ring slot batch % NBR, as row     */
/* pairs: [pair][lane][2] float2, so that ds_read_b128 gives a lane two adjacent rows.           */
/* Zero taps meet rows outside an output's own window: exact for finite samples only, so the     */
/* loader looks at every value it brings in and a workgroup that has seen an infinity or a NaN   */
/* takes the literal loop (per-row tests) for the rest of its segment.                           */
/* ------------------------------------------------------------------------------------------ */
constexpr int RSR_ROWS = 96;    // rows a workgroup can hold in registers for the next step
constexpr int RSR_HEAD = 4 + 4; // ints per group header: top batch, batches, ring offset, -, pidx[R]

template <int R>
__device__ __forceinline__ void rs_walk_asm(fmd_f2v (&acc)[R], unsigned& off, unsigned& cnt, unsigned lane16,
                                            unsigned wrap, unsigned klo, unsigned khi, unsigned kinc);
template <>
__device__ __forceinline__ void rs_walk_asm<4>(fmd_f2v (&acc)[4], unsigned& off, unsigned& cnt, unsigned lane16,
                                               unsigned wrap, unsigned klo, unsigned khi, unsigned kinc)
{
#include "fmd_rs_walk_r4.inc"
}
template <>
__device__ __forceinline__ void rs_walk_asm<2>(fmd_f2v (&acc)[2], unsigned& off, unsigned& cnt, unsigned lane16,
                                               unsigned wrap, unsigned klo, unsigned khi, unsigned kinc)
{
#include "fmd_rs_walk_r2.inc"
}

template <int R, int NW>
__device__ __forceinline__ void rs_warm_asm(unsigned tlo, unsigned thi, unsigned gstride, unsigned rounds,
                                            unsigned pace);
template <>
__device__ __forceinline__ void rs_warm_asm<4, 4>(unsigned tlo, unsigned thi, unsigned gstride, unsigned rounds,
                                                  unsigned pace)
{
#include "fmd_rs_warm_r4w4.inc"
}
template <>
__device__ __forceinline__ void rs_warm_asm<2, 8>(unsigned tlo, unsigned thi, unsigned gstride, unsigned rounds,
                                                  unsigned pace)
{
#include "fmd_rs_warm_r2w8.inc"
}
template <>
__device__ __forceinline__ void rs_warm_asm<2, 4>(unsigned tlo, unsigned thi, unsigned gstride, unsigned rounds,
                                                  unsigned pace)
{
#include "fmd_rs_warm_r2w4.inc"
}

/* The call's plan, one block per group of R outputs (positions are batch-uniform): header, the
 * group's taps by (batch, row, output) with zeros outside each output's window, and per step the
 * batches its NW groups touch.  pf / pi / k0 / k1 as in k_rs_table (DownConvert.cpp:205-212). */
template <int R, int RSR_NW>
__global__ __launch_bounds__(64) void k_rs_plan(const float* __restrict__ coeff, unsigned order, float p,
                                                float pstep, unsigned A, int RB, unsigned NBR,
                                                float* __restrict__ tab, unsigned nbm,
                                                int* __restrict__ head, int* __restrict__ steptab)
{
  const unsigned g = blockIdx.x;
  auto pidx_of = [&](unsigned i) { return (int)(p + (float)i * pstep); };
  auto extent = [&](unsigned gg, int& top, int& nb) { // batches of group gg, top one first, an even count
    const unsigned i0 = gg * R;
    if (i0 >= A)
    {
      top = 0;
      nb = 0;
      return;
    }
    const unsigned il = min(i0 + R - 1, A - 1);
    top = (pidx_of(il) + RB) >> 3;
    const int bot = (pidx_of(i0) - (int)order + RB) >> 3;
    nb = top - bot + 1;
    nb += nb & 1;
  };
  int top, nb;
  extent(g, top, nb);
  float k0[R], k1[R];
  int pi[R];
#pragma unroll
  for (int r = 0; r < R; r++)
  {
    const unsigned i = min(g * R + r, A - 1);
    const float pf = p + (float)i * pstep;
    pi[r] = (int)pf;
    k1[r] = pf - (float)pi[r];
    k0[r] = 1 - k1[r];
  }
  for (int idx = threadIdx.x; idx < nb * 8 * R; idx += 64)
  {
    const int b = idx / (8 * R), q = (idx / R) & 7, r = idx % R;
    const int row = (top - b) * 8 + 7 - q - RB; // call row
    float v = 0.0f;
#pragma unroll
    for (int rr = 0; rr < R; rr++)
      if (rr == r)
      {
        const int j = pi[rr] - row;
        if (j >= 0 && j <= (int)order)
          v = coeff[j] * k0[rr] + coeff[j + 1] * k1[rr];
      }
    tab[((size_t)g * nbm + b) * (8 * R) + (idx % (8 * R))] = v;
  }
  if (threadIdx.x == 0)
  {
    int* h = head + (size_t)g * RSR_HEAD;
    h[0] = top;
    h[1] = nb;
    h[2] = (int)(((unsigned)top % NBR) * 4096u);
    h[3] = 0;
#pragma unroll
    for (int r = 0; r < R; r++)
      h[4 + r] = pi[r];
    if (g % RSR_NW == 0)
    { // the step's batches: [bot, top] over its NW groups (the last non-empty group has the top)
      int stop = 0, sbot = 0x7fffffff;
      for (unsigned w = 0; w < (unsigned)RSR_NW; w++)
      {
        int t, n;
        extent(g + w, t, n);
        if (n > 0)
        {
          stop = t;
          sbot = min(sbot, t - n + 1);
        }
      }
      steptab[2 * (g / RSR_NW)] = stop;
      steptab[2 * (g / RSR_NW) + 1] = sbot;
    }
  }
}

template <int R, int RSR_NW>
__global__ __launch_bounds__(64 * (RSR_NW + 1)) void k_resample_ring(
    const float2* __restrict__ br, unsigned Hbb, int RB, unsigned order, const float* __restrict__ tab,
    unsigned nbm, const int* __restrict__ head, const int* __restrict__ steptab, unsigned nsteps,
    unsigned steps_per_wg, unsigned NBR, unsigned A, float2* __restrict__ out, unsigned Hout, unsigned C,
    unsigned CP, unsigned exp, unsigned pace)
{
  constexpr int RSR_PRE = RSR_ROWS / RSR_NW; // rows a wave can hold for the next step
  constexpr int LEAD = 3;                    // batches the tap warmer runs ahead of the walk
  extern __shared__ __align__(16) unsigned char rsr_smem[]; // the ring: [NBR][4 pairs][64 lanes][2 rows] float2
  __shared__ unsigned nonfinite_s;
  const unsigned lane = threadIdx.x;
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  /* The steps of all channel groups in one sequence (group-major), an equal run of it per workgroup: the
   * grid is as many workgroups as CUs are free, whatever the number of groups -- a workgroup takes a whole
   * CU's LDS, so a grid of groups x segments ran in rounds, and one CU that was not free at the start cost
   * a whole round more (0.53 instead of 0.36 ms inside the pipeline with 384 workgroups for 192 CUs).  A
   * run that crosses into the next group starts that group's ring afresh, like a segment. */
  wave_prio((exp >> 8) & 3u);
  const unsigned units = (CP / 64u) * nsteps;
  unsigned u0 = min(units, blockIdx.x * steps_per_wg);
  const unsigned u1 = min(units, u0 + steps_per_wg);
  if (u0 >= u1)
    return;
  unsigned c = 0, s0 = 0, s1 = 0, lane_off = 0;
  bool live = false;
  const unsigned ring_pairs = NBR * 4;
  // absolute row rr in memory: wave-uniform row pointer + 32-bit lane offset; and in the ring
  const char* const gbase = reinterpret_cast<const char*>(br + ((ptrdiff_t)Hbb - (ptrdiff_t)RB) * (ptrdiff_t)CP);
  const size_t row_bytes = (size_t)CP * sizeof(float2);
  // a value is finite iff its exponent field is not all ones: the largest magnitude word seen decides
  unsigned emax = 0u;
  auto look = [&](float2 v) {
    emax = max(emax, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu));
  };
  auto ring_addr = [&](unsigned pair_slot, unsigned odd) {
    return reinterpret_cast<float2*>(rsr_smem + (size_t)pair_slot * 1024 + lane * 16 + odd * 8);
  };
  // (live: padding lanes hold whatever: they must not trip the non-finite flag)
  // this wave's share of the rows r0, r0 + 1, ... r0 + n_rows - 1 (r0 a multiple of 8): rows r0 + w + NW n,
  // fetched into registers with all loads in flight, and put into the ring later
  float2 pre[RSR_PRE];
  auto fetch = [&](int r0, int mine) {
    const char* rp = gbase + (size_t)(r0 + (int)w) * row_bytes;
#pragma unroll
    for (int n = 0; n < RSR_PRE; n++)
      if (n < mine)
      { // read once: must not push the tap table out of the L2
        if (exp & 32u)
          pre[n] = *reinterpret_cast<const float2*>(rp + lane_off);
        else
        {
          const fmd_f2v v = __builtin_nontemporal_load(reinterpret_cast<const fmd_f2v*>(rp + lane_off));
          pre[n] = make_float2(v.x, v.y);
        }
        rp += RSR_NW * row_bytes;
      }
  };
  auto stash = [&](int r0, int mine) {
    unsigned ps = ((unsigned)(r0 + (int)w) >> 1) % ring_pairs; // pairs 2 apart, the row's parity is w's
#pragma unroll
    for (int n = 0; n < RSR_PRE; n++)
      if (n < mine)
      {
        look(pre[n]);
        *ring_addr(ps, w & 1u) = pre[n];
        ps += RSR_NW / 2;
        ps = ps >= ring_pairs ? ps - ring_pairs : ps;
      }
    if (emax >= 0x7f800000u && live)
      nonfinite_s = 1u;
  };
  auto share = [&](int n_rows) { return (n_rows - (int)w + RSR_NW - 1) / RSR_NW; };
  /* Wave NW computes nothing: it keeps the taps the other waves are about to load in the CU's scalar
   * cache.  Their scalar loads run one batch ahead of the arithmetic (all a wave can afford: every wait
   * is lgkmcnt(0), which also waits for whatever else it has in flight), a table line is used once, and
   * a miss takes about two batches.  The warmer touches the lines LEAD batches ahead of where the walk
   * should be, at the walk's pace (it and the walk start a step at the same barrier). */
  const bool warmer = w == (unsigned)RSR_NW;
  auto warm = [&](unsigned step, int b0, int rounds) { // batches b0 .. b0 + rounds - 1 of every group of the step
    if (rounds <= 0)
      return;
    const uint64_t ta = reinterpret_cast<uint64_t>(tab) + ((uint64_t)step * RSR_NW * nbm + (uint64_t)b0) * (32 * R);
    rs_warm_asm<R, RSR_NW>((unsigned)ta, (unsigned)(ta >> 32), nbm * (32u * R), (unsigned)rounds, pace);
  };
  if (!(exp & 64u))
  { /* The taps arrive by scalar loads one batch ahead; a load that misses the L2 (the plan kernel wrote the
     * table on some other XCD) takes longer than that.  So the workgroups of an XCD (equal blockIdx.x % 8
     * under round-robin placement: speed only) first read the table (1.7 MB for 82 steps) through the
     * vector path, an equal part each, which leaves it in their L2. */
    const size_t gsz = (size_t)nbm * (8 * R) * sizeof(float);
    const char* t0 = reinterpret_cast<const char*>(tab);
    const size_t bytes = (size_t)nsteps * RSR_NW * gsz;
    const unsigned nx = (gridDim.x + 7u) / 8u, part = blockIdx.x / 8u;
    const size_t per = ((bytes + nx - 1) / nx + 15) & ~(size_t)15;
    const size_t lo = min(bytes, part * per), hi = min(bytes, lo + per);
    unsigned sink = 0;
    for (size_t o = lo + (size_t)(threadIdx.y * 64 + lane) * 16; o + 16 <= hi; o += (size_t)(RSR_NW + 1) * 64 * 16)
    {
      const uint4 v = *reinterpret_cast<const uint4*>(t0 + o);
      sink |= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (sink == 0x7fc12345u) // never (keeps the loads)
      nonfinite_s = sink;
  }
  for (; u0 < u1; u0 += s1 - s0)
  {
  const unsigned grp = u0 / nsteps;
  s0 = u0 - grp * nsteps;
  s1 = min(nsteps, s0 + (u1 - u0));
  c = grp * 64 + lane;
  lane_off = c * (unsigned)sizeof(float2);
  live = c < C;
  emax = 0u;
  __syncthreads(); // the previous run's last step is through with the ring and the flag
  if (threadIdx.x == 0 && threadIdx.y == 0)
    nonfinite_s = 0u;
  __syncthreads();
  int topb = steptab[2 * s0];
  { // the first step's whole window
    const int r_hi = topb * 8 + 7;
    for (int r0 = steptab[2 * s0 + 1] * 8; r0 <= r_hi && !(exp & 16u); r0 += RSR_NW * RSR_PRE)
    {
      const int mine = warmer ? 0 : share(min(r_hi + 1 - r0, RSR_NW * RSR_PRE));
      fetch(r0, mine);
      stash(r0, mine);
    }
    if (warmer && !(exp & 128u))
      warm(s0, 0, LEAD);
  }
  __syncthreads();
  for (unsigned s = s0; s < s1; s++)
  {
    // rows of the next step: in flight during this step's arithmetic
    const int ntop = s + 1 < s1 ? steptab[2 * (s + 1)] : topb;
    const int r_new0 = topb * 8 + 8;
    const int mine = (exp & 4u) || warmer ? 0 : share((ntop - topb) * 8);
    fetch(r_new0, mine);
    if (warmer)
    {
      if (!(exp & 128u))
      {
        warm(s, LEAD, (int)nbm - 1 - LEAD);
        if (s + 1 < s1)
          warm(s + 1, 0, LEAD);
      }
      lds_barrier();
      topb = ntop;
      lds_barrier();
      continue;
    }
    const unsigned g = s * RSR_NW + w;
    const int* __restrict__ h = head + (size_t)g * RSR_HEAD;
    const int gtop = h[0], nb = h[1];
    fmd_f2v acc[R];
#pragma unroll
    for (int r = 0; r < R; r++)
      acc[r] = fmd_f2v{0.0f, 0.0f};
    if (nb > 0 && !(exp & 8u))
    {
      const float* __restrict__ kp = tab + (size_t)g * nbm * (8 * R);
      if (__builtin_expect(nonfinite_s == 0u, 1))
      {
        unsigned off = (unsigned)h[2], cnt = (unsigned)nb >> 1;
        const uint64_t ka = reinterpret_cast<uint64_t>(kp);
        // (the low half of a generic LDS pointer is the LDS byte address)
        rs_walk_asm<R>(acc, off, cnt, (unsigned)(size_t)rsr_smem + lane * 16u, (NBR - 1u) * 4096u, (unsigned)ka,
                       (unsigned)(ka >> 32), (exp & 1u) ? 0u : 32u * R);
      }
      else
      { // literal: only the rows of an output's own window, j ascending
        for (int b = 0; b < nb; b++)
        {
          const unsigned slot = (unsigned)(gtop - b) % NBR;
#pragma unroll
          for (int q = 0; q < 8; q++)
          {
            const int rowb = 7 - q;
            const float2 x = *ring_addr(slot * 4 + (unsigned)(rowb >> 1), (unsigned)rowb & 1u);
            const int row = (gtop - b) * 8 + rowb - RB;
#pragma unroll
            for (int r = 0; r < R; r++)
            {
              const int j = h[4 + r] - row;
              if (j >= 0 && j <= (int)order)
              {
                const float k = kp[(size_t)b * (8 * R) + q * R + r];
                acc[r].x += k * x.x;
                acc[r].y += k * x.y;
              }
            }
          }
        }
      }
      if (live && !(exp & 2u))
      {
#pragma unroll
        for (int r = 0; r < R; r++)
          if (g * R + r < A) // (stereo, mono) = ProcessTwo's (A, B): x came from baseband -> mono
            out[(size_t)(Hout + g * R + r) * CP + c] = make_float2(acc[r].y, acc[r].x);
      }
    }
    lds_barrier(); // every wave is done with this step's rows
    stash(r_new0, mine);
    topb = ntop;
    lds_barrier();
  }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* K8: audio tail, one lane per channel: ProcessDeemphasisFilter (FmDecode.cpp:348-359),        */
/*     19 kHz notch cIirFilter::ProcessTwo (IirFilter.cpp:89-105), L/R matrix (:473-499).       */
/* ------------------------------------------------------------------------------------------ */
constexpr int AT_STEPS = 16; // rows in flight per lane (32: slower inside the pipeline, 0.83 against 0.65 ms)

/* One lane per channel.  The channel-major output ([C][stride], what ProcessStream's caller gets) is
 * written by every lane into its own channel's row, one frame (8 B) per store: the 16 stores that
 * fill a 128-byte line follow each other within ~2000 cycles and meet in the L2.  (Until round 4 the
 * frames went through an LDS tile for 64-byte segments per store; the tile's 8.7 KB kept the
 * whole-CU resampler off every CU an audio tail was on, and the stores are not what bounds a
 * lane-per-channel recurrence.)  The status record goes to device memory; k_status_publish takes it
 * to the host. */
__global__ __launch_bounds__(64) void k_audio_tail(const float2* __restrict__ lp, unsigned A,
                                                   unsigned C, unsigned CP, AudioConsts k,
                                                   ChannelState st, float* __restrict__ audio,
                                                   size_t audio_stride, unsigned stereo_q,
                                                   unsigned call_index)
{
  __builtin_amdgcn_s_setprio(3);
  const unsigned lane = threadIdx.x;
  const unsigned c0 = blockIdx.x * 64 + lane;
  const bool active = c0 < C;
  const unsigned c = active ? c0 : C - 1;
  float de_re = st.F(F_DE_RE)[c], de_im = st.F(F_DE_IM)[c];
  float w1a = st.F(F_N_W1A)[c], w2a = st.F(F_N_W2A)[c], w1b = st.F(F_N_W1B)[c], w2b = st.F(F_N_W2B)[c];
  const int stereo = st.I(I_STEREO_Q0 + (int)stereo_q)[c];
  const float one_minus_alpha = 1.0f - k.de_alpha;
  // cRadioReceiver::SamplesMeanRMS over the packet (RadioReceiver.cpp:584-598): float sums over
  // the interleaved samples L0, R0, L1, R1, ... in that order
  float vsum = 0.0f, vsumsq = 0.0f;

  auto frame = [&](float2 v) -> float2 { // v.x = stereo, v.y = mono (ProcessTwo's A, B)
    de_re = one_minus_alpha * de_re + k.de_alpha * v.x;
    const float s0 = de_re * 2.0f;
    de_im = one_minus_alpha * de_im + k.de_alpha * v.y;
    const float m0 = de_im * 2.0f;
    const float w0a = s0 - k.n_a1 * w1a - k.n_a2 * w2a;
    const float w0b = m0 - k.n_a1 * w1b - k.n_a2 * w2b;
    const float s = k.n_b0 * w0a + k.n_b1 * w1a + k.n_b2 * w2a;
    const float m = k.n_b0 * w0b + k.n_b1 * w1b + k.n_b2 * w2b;
    w2a = w1a;
    w1a = w0a;
    w2b = w1b;
    w1b = w0b;
    const float mm = m * 0.5f;
    const float2 o = stereo ? make_float2((m + s) * 0.5f, (m - s) * 0.5f) : make_float2(mm, mm);
    vsum += o.x;
    vsumsq += o.x * o.x;
    vsum += o.y;
    vsumsq += o.y * o.y;
    return o;
  };
  float2* __restrict__ o = reinterpret_cast<float2*>(audio + (size_t)c * audio_stride);

  unsigned i0 = 0;
  // full tiles: the loads of the next tile are in flight while this one goes through the recurrence
  // out of registers (past the last full tile: clamped rows nobody uses)
  float2 vnext[AT_STEPS];
#pragma unroll
  for (unsigned u = 0; u < AT_STEPS; u++)
    vnext[u] = lp[(size_t)min(u, A - 1) * CP + c];
  for (; i0 + AT_STEPS <= A; i0 += AT_STEPS)
  {
    float2 vin[AT_STEPS];
#pragma unroll
    for (unsigned u = 0; u < AT_STEPS; u++)
      vin[u] = vnext[u];
#pragma unroll
    for (unsigned u = 0; u < AT_STEPS; u++)
      vnext[u] = lp[(size_t)min(i0 + AT_STEPS + u, A - 1) * CP + c];
#pragma unroll
    for (unsigned u = 0; u < AT_STEPS; u++)
    {
      const float2 f = frame(vin[u]);
      if (active)
        o[i0 + u] = f;
    }
  }
  if (i0 < A)
  {
    const unsigned cnt = A - i0;
#pragma unroll
    for (unsigned u = 0; u < AT_STEPS; u++) // the ragged last tile is already in vnext
      if (u < cnt)
      {
        const float2 f = frame(vnext[u]);
        if (active)
          o[i0 + u] = f;
      }
  }
  if (active)
  {
    st.F(F_DE_RE)[c] = de_re;
    st.F(F_DE_IM)[c] = de_im;
    st.F(F_N_W1A)[c] = w1a;
    st.F(F_N_W2A)[c] = w2a;
    st.F(F_N_W1B)[c] = w1b;
    st.F(F_N_W2B)[c] = w2b;
    // mean = vsum / n, rms = sqrt(vsumsq / n) in float (n = floats in the packet), then
    // m_AudioLevel = 0.95 * m_AudioLevel + 0.05 * audio_rms in double (RadioReceiver.cpp:526-528)
    const float n = (float)(2u * A);
    const float rms = sqrtf(vsumsq / n);
    const float mean = vsum / n;
    const float level = (float)(0.95 * (double)st.F(F_AUDIO_LEVEL)[c] + 0.05 * (double)rms);
    st.F(F_AUDIO_MEAN)[c] = mean;
    st.F(F_AUDIO_RMS)[c] = rms;
    st.F(F_AUDIO_LEVEL)[c] = level;
    /* The call is complete for this channel: its status record (see HostStatusWord).  The level
     * meters are the state arrays as they stand now; the stereo flag is this call's own copy.  With
     * overlapped calls (concurrency 2) the next call's IF / baseband meters may already be in -- the
     * reference's status thread reads its decoder mid-call too (RadioReceiver.cpp:544-572 against
     * :524, no common lock). */
    unsigned* __restrict__ h = st.ds + c;
    const size_t CPs = st.CP;
    h[HS_IF_LEVEL * CPs] = __float_as_uint(st.F(F_IF_LEVEL)[c]);
    h[HS_BB_MEAN * CPs] = __float_as_uint(st.F(F_BB_MEAN)[c]);
    h[HS_BB_LEVEL * CPs] = __float_as_uint(st.F(F_BB_LEVEL)[c]);
    h[HS_P_LEVEL * CPs] = __float_as_uint(st.F(F_P_LEVEL)[c]);
    h[HS_STEREO * CPs] = (unsigned)stereo;
    h[HS_AUDIO_MEAN * CPs] = __float_as_uint(mean);
    h[HS_AUDIO_RMS * CPs] = __float_as_uint(rms);
    h[HS_AUDIO_LEVEL * CPs] = __float_as_uint(level);
  }
}

/* K7 + K8 as one kernel: the 29-tap audio low-pass (cFirFilter::ProcessTwo, FirFilter.cpp:387-413) in front of
 * the tail above, one lane per channel, nothing between them in memory.  The filter's delay line is the
 * reference's own ring buffer m_cZBuf[0 .. T-1], held in REGISTERS: position j keeps its sample until the ring
 * index m_State comes round again, and an output adds H[T - m_State + j] * Z[j] for j = 0 .. T-1 in that order --
 * with a0 = (T - m_State) % T (host-tracked: (g0 + i) % T for output i) that is tap (a0 + j) % T, age a0 first.
 *
 * Rounds of T outputs with a0 = 0 .. T-1 are straight-line code: every ring position and every tap a compile-time
 * register (taps in scalar register pairs, picked by op_sel), the multiply-add chain of every phase one asm block
 * (fmd_alp_mac.inc, tools/gen_audio_lpf_asm.py) -- each product consumed two instructions after it was issued:
 * 58 issue slots per frame and no wait slots (the compiler's own order costs 87, and it pads every inline-asm
 * statement it cannot see into).  A frame's input was loaded a whole round earlier.  The frames in front of the first whole round and behind
 * the last one (a call starts at a0 = g0 % T) go through a generic body: the new sample enters the ring by a
 * dynamic register index, the taps are read from a doubled table at a0.  (A first version dispatched every frame
 * to one of T bodies by a switch: correct, but the compiler cannot count outstanding loads across the switch and
 * waited for ALL of them in every frame -- 1570 cycles per frame alone on the chip.)
 * The delay line's T - 1 history rows are read from the front of the resampler's output buffer and written to
 * the front of the other parity's at the end, where k_ring_fir4's roll kept them: the two forms may follow each
 * other from call to call.
 * What it saves: the low-pass output's round trip (0.15 GB per call at 8192 channels) and a time-parallel
 * kernel of 10 000 small workgroups that ran starved beside the IF FIR (0.68 ms inside the pipeline for 0.075
 * alone); what it costs: 58 more packed instructions per frame in a lane-per-channel recurrence. */
typedef float fmd_f32v __attribute__((ext_vector_type(32)));
typedef float fmd_f16v __attribute__((ext_vector_type(16)));
template <int U>
__device__ __forceinline__ void alp_mac(fmd_f2v& acc, const fmd_f32v& zl, const fmd_f32v& zh, const fmd_f16v& ta,
                                        const fmd_f16v& tb);
#include "fmd_alp_mac.inc"

template <int T>
struct AudioLpfTail
{
  static_assert(T == 29, "ring positions 0 .. 14 in zl, 15 .. 28 in zh, pair 15 of each a scratch slot");
  fmd_f32v zl, zh; // the ring: position p < 15 at zl[2p .. 2p+1], p >= 15 at zh[2(p-15) ..]; floats 30, 31: scratch
  fmd_f16v ta, tb; // taps 0 .. 15 and 16 .. 28, wave-uniform: scalar registers (fmd_alp_mac.inc names them)
  float de_re, de_im, w1a, w2a, w1b, w2b, vsum, vsumsq, one_minus_alpha;
  int stereo;
  AudioConsts k;
  const float2* __restrict__ rs;
  const float* __restrict__ taps2; // taps twice in a row: taps2[a0 + j] = tap (a0 + j) % T
  float2* __restrict__ o;
  size_t cp;
  unsigned c, A, i;
  bool active;

  template <int P>
  __device__ __forceinline__ fmd_f2v getz() const
  {
    if constexpr (P < 15)
      return __builtin_shufflevector(zl, zl, 2 * P, 2 * P + 1);
    else
      return __builtin_shufflevector(zh, zh, 2 * (P - 15), 2 * (P - 15) + 1);
  }
  template <int P>
  __device__ __forceinline__ void setz(float2 v)
  {
    if constexpr (P < 15)
    {
      zl[2 * P] = v.x;
      zl[2 * P + 1] = v.y;
    }
    else
    {
      zh[2 * (P - 15)] = v.x;
      zh[2 * (P - 15) + 1] = v.y;
    }
  }

  __device__ __forceinline__ float2 frame(float2 v)
  { // v.x = stereo, v.y = mono (ProcessTwo's A, B): the statements of k_audio_tail
    de_re = one_minus_alpha * de_re + k.de_alpha * v.x;
    const float s0 = de_re * 2.0f;
    de_im = one_minus_alpha * de_im + k.de_alpha * v.y;
    const float m0 = de_im * 2.0f;
    const float w0a = s0 - k.n_a1 * w1a - k.n_a2 * w2a;
    const float w0b = m0 - k.n_a1 * w1b - k.n_a2 * w2b;
    const float s = k.n_b0 * w0a + k.n_b1 * w1a + k.n_b2 * w2a;
    const float m = k.n_b0 * w0b + k.n_b1 * w1b + k.n_b2 * w2b;
    w2a = w1a;
    w1a = w0a;
    w2b = w1b;
    w1b = w0b;
    const float mm = m * 0.5f;
    const float2 f = stereo ? make_float2((m + s) * 0.5f, (m - s) * 0.5f) : make_float2(mm, mm);
    vsum += f.x;
    vsumsq += f.x * f.x;
    vsum += f.y;
    vsumsq += f.y * f.y;
    return f;
  }

  template <int U>
  __device__ __forceinline__ void slot(float2 x)
  { // the output whose a0 is U: m_State = (T - U) % T takes the new sample
    setz<(T - U) % T>(x);
    fmd_f2v acc;
    alp_mac<U>(acc, zl, zh, ta, tb);
    const float2 f = frame(make_float2(acc.x, acc.y));
    if (active)
      o[i] = f;
    i++;
  }
  template <int U>
  __device__ __forceinline__ void round(float2 (&nxt)[T])
  {
    if constexpr (U < T)
    {
      const float2 x = nxt[U];
      nxt[U] = row(i + (unsigned)T); // this slot's input of the next round
      slot<U>(x);
      round<U + 1>(nxt);
    }
  }

  // any a0: the new sample by a dynamic register index (both halves of the ring take it, the one it does
  // not belong to into its scratch slot), the taps from the doubled table
  __device__ __forceinline__ void generic(unsigned a0, float2 x)
  {
    const unsigned su = a0 ? (unsigned)T - a0 : 0u;
    const unsigned il = su < 15u ? 2u * su : 30u, ih = su >= 15u ? 2u * (su - 15u) : 30u;
    zl[il] = x.x;
    zl[il + 1] = x.y;
    zh[ih] = x.x;
    zh[ih + 1] = x.y;
    const float* __restrict__ tb = taps2 + a0;
    float2 acc = make_float2(tb[0] * zl[0], tb[0] * zl[1]);
#pragma unroll
    for (int j = 1; j < 15; j++)
    {
      acc.x += tb[j] * zl[2 * j];
      acc.y += tb[j] * zl[2 * j + 1];
    }
#pragma unroll
    for (int j = 15; j < T; j++)
    {
      acc.x += tb[j] * zh[2 * (j - 15)];
      acc.y += tb[j] * zh[2 * (j - 15) + 1];
    }
    const float2 f = frame(acc);
    if (active)
      o[i] = f;
    i++;
  }
  __device__ __forceinline__ float2 row(unsigned t) const // input of output t (clamped: nobody uses the rest)
  {
    return rs[(size_t)((unsigned)T - 1u + min(t, A - 1u)) * cp + c];
  }
  // outputs i .. stop - 1 through the generic body, inputs four frames ahead
  __device__ __forceinline__ void stretch(unsigned& a0, unsigned stop)
  {
    float2 x0 = row(i), x1 = row(i + 1), x2 = row(i + 2), x3 = row(i + 3);
    while (i < stop)
    {
      const float2 x = x0;
      x0 = x1;
      x1 = x2;
      x2 = x3;
      x3 = row(i + 4);
      generic(a0, x);
      a0 = a0 + 1u == (unsigned)T ? 0u : a0 + 1u;
    }
  }
};

__global__ __launch_bounds__(64) void k_audio_lpf_tail29(const float2* __restrict__ rs, float2* __restrict__ rs_next,
                                                         unsigned A, unsigned g0, const float* __restrict__ taps2,
                                                         unsigned C, unsigned CP, AudioConsts k, ChannelState st,
                                                         float* __restrict__ audio, size_t audio_stride,
                                                         unsigned stereo_q, unsigned call_index, unsigned prio)
{
  constexpr int T = 29;
  wave_prio(prio);
  const unsigned lane = threadIdx.x;
  const unsigned c0 = blockIdx.x * 64 + lane;
  AudioLpfTail<T> s;
  s.active = c0 < C;
  const unsigned c = s.active ? c0 : C - 1;
  s.c = c;
  s.cp = CP;
  s.A = A;
  s.i = 0;
  s.k = k;
  s.rs = rs;
  s.taps2 = taps2;
  s.de_re = st.F(F_DE_RE)[c];
  s.de_im = st.F(F_DE_IM)[c];
  s.w1a = st.F(F_N_W1A)[c];
  s.w2a = st.F(F_N_W2A)[c];
  s.w1b = st.F(F_N_W1B)[c];
  s.w2b = st.F(F_N_W2B)[c];
  s.stereo = st.I(I_STEREO_Q0 + (int)stereo_q)[c];
  s.one_minus_alpha = 1.0f - k.de_alpha;
  s.vsum = 0.0f;
  s.vsumsq = 0.0f;
  s.o = reinterpret_cast<float2*>(audio + (size_t)c * audio_stride);
#pragma unroll
  for (int m = 0; m < 16; m++)
  {
    s.ta[m] = taps2[m];
    s.tb[m] = taps2[16 + m]; // (taps 29 .. 31 of the doubled table: unused)
  }
  const unsigned a0s = g0 % (unsigned)T;
  // ring position j holds the sample of age (a0s + j) % T before the first output (buffer row of time t is
  // T - 1 + t); age 0 is the position the first output overwrites
  s.zl = 0.0f;
  s.zh = 0.0f;
  {
    float2 h[T];
#pragma unroll
    for (int j = 0; j < T; j++)
    {
      unsigned age = a0s + (unsigned)j;
      age = age >= (unsigned)T ? age - (unsigned)T : age;
      h[j] = rs[(size_t)((unsigned)T - 1u - max(age, 1u)) * CP + c];
    }
#pragma unroll
    for (int j = 0; j < 15; j++)
    {
      s.zl[2 * j] = h[j].x;
      s.zl[2 * j + 1] = h[j].y;
    }
#pragma unroll
    for (int j = 15; j < T; j++)
    {
      s.zh[2 * (j - 15)] = h[j].x;
      s.zh[2 * (j - 15) + 1] = h[j].y;
    }
  }
  unsigned a0 = a0s;
  // the frames in front of the first whole round
  s.stretch(a0, min(A, a0s ? (unsigned)T - a0s : 0u));
  if (s.i + (unsigned)T <= A)
  { // whole rounds: a0 = 0 .. T - 1, every frame's input loaded a round ahead
    float2 nxt[T];
#pragma unroll
    for (int u = 0; u < T; u++)
      nxt[u] = s.row(s.i + (unsigned)u);
    while (s.i + (unsigned)T <= A)
      s.template round<0>(nxt);
  }
  s.stretch(a0, A); // (a0 is 0 here unless the call had no whole round)
  if (s.active)
  {
    // the delay line for the next call, where k_ring_fir4's roll keeps it: row T - 1 - age of the other buffer
    const unsigned a0e = (a0s + A) % (unsigned)T;
    float2 h[T];
#pragma unroll
    for (int j = 0; j < 15; j++)
      h[j] = make_float2(s.zl[2 * j], s.zl[2 * j + 1]);
#pragma unroll
    for (int j = 15; j < T; j++)
      h[j] = make_float2(s.zh[2 * (j - 15)], s.zh[2 * (j - 15) + 1]);
#pragma unroll
    for (int j = 0; j < T; j++)
    {
      unsigned age = a0e + (unsigned)j;
      age = age >= (unsigned)T ? age - (unsigned)T : age;
      if (age != 0)
        rs_next[(size_t)((unsigned)T - 1u - age) * CP + c] = h[j];
    }
    st.F(F_DE_RE)[c] = s.de_re;
    st.F(F_DE_IM)[c] = s.de_im;
    st.F(F_N_W1A)[c] = s.w1a;
    st.F(F_N_W2A)[c] = s.w2a;
    st.F(F_N_W1B)[c] = s.w1b;
    st.F(F_N_W2B)[c] = s.w2b;
    const float n = (float)(2u * A);
    const float rms = sqrtf(s.vsumsq / n);
    const float mean = s.vsum / n;
    const float level = (float)(0.95 * (double)st.F(F_AUDIO_LEVEL)[c] + 0.05 * (double)rms);
    st.F(F_AUDIO_MEAN)[c] = mean;
    st.F(F_AUDIO_RMS)[c] = rms;
    st.F(F_AUDIO_LEVEL)[c] = level;
    unsigned* __restrict__ h2 = st.ds + c; // the call's status record (see k_audio_tail)
    const size_t CPs = st.CP;
    h2[HS_IF_LEVEL * CPs] = __float_as_uint(st.F(F_IF_LEVEL)[c]);
    h2[HS_BB_MEAN * CPs] = __float_as_uint(st.F(F_BB_MEAN)[c]);
    h2[HS_BB_LEVEL * CPs] = __float_as_uint(st.F(F_BB_LEVEL)[c]);
    h2[HS_P_LEVEL * CPs] = __float_as_uint(st.F(F_P_LEVEL)[c]);
    h2[HS_STEREO * CPs] = (unsigned)s.stereo;
    h2[HS_AUDIO_MEAN * CPs] = __float_as_uint(mean);
    h2[HS_AUDIO_RMS * CPs] = __float_as_uint(rms);
    h2[HS_AUDIO_LEVEL * CPs] = __float_as_uint(level);
  }
}

/* The last kernel of a call: every channel's status record from device memory to the host's snapshot
 * under the per-channel sequence lock (HostStatusWord), a thread per channel -- one kernel of a few
 * waves pays the two system-scope fences, not the latency-bound audio tail. */
__global__ __launch_bounds__(256) void k_status_publish(ChannelState st, unsigned C, unsigned call_index)
{
  const unsigned c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C)
    return;
  const size_t CPs = st.CP;
  unsigned v[HS_WORDS];
#pragma unroll
  for (int w = HS_SEQ_BEGIN + 1; w < HS_SEQ_END; w++)
    v[w] = st.ds[(size_t)w * CPs + c];
  volatile unsigned* h = st.hs + c;
  h[HS_SEQ_BEGIN * CPs] = call_index;
  __threadfence_system();
#pragma unroll
  for (int w = HS_SEQ_BEGIN + 1; w < HS_SEQ_END; w++)
    h[(size_t)w * CPs] = v[w];
  __threadfence_system();
  h[HS_SEQ_END * CPs] = call_index;
}

/* ------------------------------------------------------------------------------------------ */
/* RDS groups of one call's queue -> fixed-size records in device memory (the N > 1 gather of    */
/* bench.py sends them to rank 0 as they are: no host round trip).  Row = 4 x int32:            */
/* channel + 1 + channel_offset, call index, b0 | b1 << 16, b2 | b3 << 16; rows nobody writes    */
/* stay zero (the host zeroes the buffer first).  One workgroup per queue; `cursor` is the      */
/* running row count over the queues drained into the same buffer.  Empties the queue.          */
/* ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_rds_export(const RdsGroupRec* __restrict__ queue,
                                                    unsigned* __restrict__ queue_count, unsigned queue_cap,
                                                    int4* __restrict__ rec, unsigned cap,
                                                    unsigned* __restrict__ cursor, unsigned channel_offset,
                                                    unsigned* __restrict__ err)
{
  __shared__ unsigned base_s;
  const unsigned n = min(*queue_count, queue_cap);
  if (threadIdx.x == 0)
    base_s = atomicAdd(cursor, n);
  __syncthreads();
  const unsigned base = base_s;
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
  {
    const RdsGroupRec r = queue[i];
    if (base + i < cap)
      rec[base + i] = make_int4((int)(r.channel + 1u + channel_offset), (int)r.call_index,
                                (int)((unsigned)r.blocks[0] | ((unsigned)r.blocks[1] << 16)),
                                (int)((unsigned)r.blocks[2] | ((unsigned)r.blocks[3] << 16)));
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    if (base + n > cap)
      dev_error(err + 1, DEVERR_RDS_QUEUE_FULL); // more groups than the caller's record buffer holds
    *queue_count = 0;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Stream probe (fmd_batch_create): one wave that stays busy for `cycles`, and a no-op.         */
/* ------------------------------------------------------------------------------------------ */
__global__ void k_probe_spin(long long cycles, int* sink)
{
  const long long t0 = __builtin_amdgcn_s_memtime();
  int n = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < cycles)
    n++;
  if (sink && n < 0)
    *sink = n;
}
/* One wave that does nothing for `ticks` of the 100 MHz clock (see the post chain's start in
 * fmd_batch.hip). */
__global__ void k_delay(unsigned ticks)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
    __builtin_amdgcn_s_sleep(4);
}

__global__ void k_probe_nop(int* sink)
{
  if (sink && threadIdx.x == 12345)
    *sink = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Test aid: the device builds of the fmd_math.h helpers on arrays of arguments, so their      */
/* device-only code (reciprocal-based division, ballot branches, table forms) can be swept      */
/* against the host libm directly (fmd_debug_math).                                             */
/* ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(64) void k_debug_math(int what, unsigned n, const float* __restrict__ a,
                                                   const float* __restrict__ b, float* __restrict__ o0,
                                                   float* __restrict__ o1,
                                                   const double* __restrict__ sctab_g, FmdSincosTab sct,
                                                   const double* __restrict__ sctab256_g)
{
  __shared__ double sctab[2 * FMD_SINCOS_TAB_SIZE];
  __shared__ double sctab256[2 * FMD_SINCOS_P256_SIZE];
  __shared__ float atab[FMD_ATAN_TAB_FLOATS];
  for (unsigned i = threadIdx.x; i < 2 * FMD_SINCOS_TAB_SIZE; i += 64)
    sctab[i] = sctab_g[i];
  for (unsigned i = threadIdx.x; i < 2 * FMD_SINCOS_P256_SIZE; i += 64)
    sctab256[i] = sctab256_g[i];
  if (threadIdx.x == 0)
    fmd_atan_table_fill(atab);
  __syncthreads();
  for (unsigned i = blockIdx.x * 64 + threadIdx.x; i < (n + 63) / 64 * 64; i += gridDim.x * 64)
  { // whole waves stay in the loop: the helpers use wave-wide ballots
    const unsigned k = min(i, n - 1);
    float r0 = 0.0f, r1 = 0.0f;
    switch (what)
    {
      case 0:
        r0 = fmd_atan2f_tab(a[k], b[k], atab);
        break;
      case 1:
        r0 = fmd_atan2f(a[k], b[k]);
        break;
      case 2:
        fmd_sincos_tab(a[k], sctab, sct, &r0, &r1);
        break;
      case 3:
        fmd_sincos_nco(a[k], &r0, &r1);
        break;
      case 4:
        r0 = fmd_div_midrange(a[k], b[k]);
        break;
      case 5:
        r0 = fmd_u8_to_f32((unsigned)a[k]);
        break;
      case 7:
        fmd_sincos_p256(a[k], sctab256, &r0, &r1);
        break;
      default:
        r0 = fmd_rds_arctan2(a[k], b[k]);
    }
    if (i < n)
    {
      o0[i] = r0;
      o1[i] = r1;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* history roll: rows [n, n+H) -> [0, H) of a time-major buffer (element size ES floats)        */
/* ------------------------------------------------------------------------------------------ */
/* dst rows [0, H) <- src rows [n, n+H): the last H rows of (history + n new rows).  src == dst
 * for single buffers (then n >= H is required for the row-parallel form), src != dst for the
 * double-buffered ones. */
/* Up to four rolls of float2 buffers in ONE launch (blockIdx.z = job): the history tails of a chain's
 * stages, all due at the chain's end.  Same semantics per job as k_roll. */
struct RollSet
{
  const float2* src[4];
  float2* dst[4];
  unsigned H[4], n[4];
};
__global__ void k_roll_set(RollSet rs, unsigned CP)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP)
    return;
  const unsigned job = blockIdx.z;
  const float2* src = rs.src[job];
  float2* dst = rs.dst[job];
  const unsigned H = rs.H[job], n = rs.n[job];
  if (n >= H || src != dst)
  {
    for (unsigned r = blockIdx.y; r < H; r += gridDim.y)
      dst[(size_t)r * CP + c] = src[(size_t)(r + n) * CP + c];
  }
  else if (blockIdx.y == 0)
  {
    for (unsigned r = 0; r < H; r++)
      dst[(size_t)r * CP + c] = src[(size_t)(r + n) * CP + c];
  }
}

template <typename T>
__global__ void k_roll(const T* src, T* dst, unsigned H, unsigned n, unsigned CP)
{
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP)
    return;
  if (n >= H || src != dst)
  { // source and destination rows are disjoint: one row per blockIdx.y
    for (unsigned r = blockIdx.y; r < H; r += gridDim.y)
      dst[(size_t)r * CP + c] = src[(size_t)(r + n) * CP + c];
  }
  else if (blockIdx.y == 0)
  { // overlapping (tiny block): ascending order is safe (destination row r < source row r + n)
    for (unsigned r = 0; r < H; r++)
      dst[(size_t)r * CP + c] = src[(size_t)(r + n) * CP + c];
  }
}

} // namespace fmd
