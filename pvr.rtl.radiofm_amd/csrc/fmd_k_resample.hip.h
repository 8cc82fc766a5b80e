/*
 * fmd_k_resample.hip.h -- the two fractional resamplers (k_rs_table + k_resample; k_rs_plan + k_resample_ring).
 * Part of fmd_kernels.hip.h (layout, numerics contract and citations: see there and fmd_k_common.hip.h).
 */
#pragma once

#include "fmd_k_common.hip.h"

namespace fmd
{

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

// (FMA: every multiply / add pair fused -- FMD_FIR_FMA_PARITY_WAIVED, the form that is measured against the
// parity mode; two outputs per wave only)
template <int R, bool FMA = false>
__device__ __forceinline__ void rs_walk_asm(fmd_f2v (&acc)[R], unsigned& off, unsigned& cnt, unsigned lane16,
                                            unsigned wrap, unsigned klo, unsigned khi, unsigned kinc);
template <>
__device__ __forceinline__ void rs_walk_asm<4, false>(fmd_f2v (&acc)[4], unsigned& off, unsigned& cnt,
                                                      unsigned lane16, unsigned wrap, unsigned klo, unsigned khi,
                                                      unsigned kinc)
{
#include "fmd_rs_walk_r4.inc"
}
template <>
__device__ __forceinline__ void rs_walk_asm<2, false>(fmd_f2v (&acc)[2], unsigned& off, unsigned& cnt,
                                                      unsigned lane16, unsigned wrap, unsigned klo, unsigned khi,
                                                      unsigned kinc)
{
#include "fmd_rs_walk_r2.inc"
}
template <>
__device__ __forceinline__ void rs_walk_asm<2, true>(fmd_f2v (&acc)[2], unsigned& off, unsigned& cnt,
                                                     unsigned lane16, unsigned wrap, unsigned klo, unsigned khi,
                                                     unsigned kinc)
{
#include "fmd_rs_walk_r2_fma.inc"
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

template <int R, int RSR_NW, bool FMA = false>
__global__ __launch_bounds__(64 * (RSR_NW + 1)) void k_resample_ring(
    const float2* __restrict__ br, unsigned Hbb, int RB, unsigned order, const float* __restrict__ tab,
    unsigned nbm, const int* __restrict__ head, const int* __restrict__ steptab, unsigned nsteps,
    unsigned steps_per_wg, unsigned NBR, unsigned A, float2* __restrict__ out, unsigned Hout, unsigned C,
    unsigned CP, unsigned prio)
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
  wave_prio(prio);
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
        const fmd_f2v v = __builtin_nontemporal_load(reinterpret_cast<const fmd_f2v*>(rp + lane_off));
        pre[n] = make_float2(v.x, v.y);
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
    rs_warm_asm<R, RSR_NW>((unsigned)ta, (unsigned)(ta >> 32), nbm * (32u * R), (unsigned)rounds, 0u);
  };
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
    for (int r0 = steptab[2 * s0 + 1] * 8; r0 <= r_hi; r0 += RSR_NW * RSR_PRE)
    {
      const int mine = warmer ? 0 : share(min(r_hi + 1 - r0, RSR_NW * RSR_PRE));
      fetch(r0, mine);
      stash(r0, mine);
    }
    if (warmer)
      warm(s0, 0, LEAD);
  }
  __syncthreads();
  for (unsigned s = s0; s < s1; s++)
  {
    // rows of the next step: in flight during this step's arithmetic
    const int ntop = s + 1 < s1 ? steptab[2 * (s + 1)] : topb;
    const int r_new0 = topb * 8 + 8;
    const int mine = warmer ? 0 : share((ntop - topb) * 8);
    fetch(r_new0, mine);
    if (warmer)
    {
      warm(s, LEAD, (int)nbm - 1 - LEAD);
      if (s + 1 < s1)
        warm(s + 1, 0, LEAD);
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
    if (nb > 0)
    {
      const float* __restrict__ kp = tab + (size_t)g * nbm * (8 * R);
      if (__builtin_expect(nonfinite_s == 0u, 1))
      {
        unsigned off = (unsigned)h[2], cnt = (unsigned)nb >> 1;
        const uint64_t ka = reinterpret_cast<uint64_t>(kp);
        // (the low half of a generic LDS pointer is the LDS byte address)
        rs_walk_asm<R, FMA>(acc, off, cnt, (unsigned)(size_t)rsr_smem + lane * 16u, (NBR - 1u) * 4096u, (unsigned)ka,
                       (unsigned)(ka >> 32), 32u * R);
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
                if (FMA)
                {
                  acc[r].x = __builtin_fmaf(k, x.x, acc[r].x);
                  acc[r].y = __builtin_fmaf(k, x.y, acc[r].y);
                }
                else
                {
                  acc[r].x += k * x.x;
                  acc[r].y += k * x.y;
                }
              }
            }
          }
        }
      }
      if (live)
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

} // namespace fmd
