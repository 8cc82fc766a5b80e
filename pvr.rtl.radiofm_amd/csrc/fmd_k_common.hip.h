/*
 * fmd_k_common.hip.h -- what every kernel file of the batched FM decoder shares: constants passed by value, the
 * per-channel state layout, device error words, LDS / priority helpers.  (fmd_kernels.hip.h includes the stages.)
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

/* Which capture channel c tunes (fmd_batch_set_channels_per_capture: cpc consecutive channels share one).
 * c is wave-uniform, but the compiler divides in vector registers: without the readfirstlane the capture's base
 * address -- and with it every input load's address arithmetic -- lives in VGPRs (config 5's IF FIR: 3.22 instead
 * of 3.0 ms alone, measured round 6). */
__device__ __forceinline__ unsigned capture_of(unsigned c, unsigned cpc)
{
  return cpc > 1u ? (unsigned)__builtin_amdgcn_readfirstlane((int)(c / cpc)) : c;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  // std::complex<float> product: (ac - bd) + i(ad + bc), four products and two sums, each rounded
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

typedef float fmd_f2v __attribute__((ext_vector_type(2)));

} // namespace fmd
