/*
 * fmd_k_tail.hip.h -- audio tail (k_audio_tail), status record, RDS record export, stream probes,
 * device-math test kernel, history rolls.
 * Part of fmd_kernels.hip.h (layout, numerics contract and citations: see there and fmd_k_common.hip.h).
 */
#pragma once

#include "fmd_k_common.hip.h"

namespace fmd
{

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
__global__ __launch_bounds__(256) void k_audio_tail(const float2* __restrict__ lp, unsigned A,
                                                   unsigned C, unsigned CP, AudioConsts k,
                                                   ChannelState st, float* __restrict__ audio,
                                                   size_t audio_stride, unsigned stereo_q,
                                                   unsigned call_index)
{
  __builtin_amdgcn_s_setprio(3);
  const unsigned lane = threadIdx.x;
  // (blockDim.y channel groups per workgroup, a wave each, nothing shared: "light_pack" -- a CU that is awake for
  // one wave draws as much as one that is busy, so the light part's waves go four to a CU: MEASUREMENTS, round 5)
  const unsigned c0 = (blockIdx.x * blockDim.y + threadIdx.y) * 64 + lane;
  if (c0 - lane >= CP)
    return;
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
