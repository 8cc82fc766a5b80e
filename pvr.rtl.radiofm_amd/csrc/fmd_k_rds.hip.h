/*
 * fmd_k_rds.hip.h -- RDS branch: half-band decimators (k_halfband*, k_halfband_chain), the ring-buffer FIR filters
 * (k_ring_fir, k_ring_fir4: RDS low-pass, matched filter, audio low-pass), RDS PLL and bit / block recovery
 * (k_rds_pll, k_rds_bits).
 * Part of fmd_kernels.hip.h (layout, numerics contract and citations: see there and fmd_k_common.hip.h).
 */
#pragma once

#include "fmd_k_common.hip.h"

namespace fmd
{

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

/* CCicN3DecimateBy2::DecBy2 (DownConvert.cpp:706-727): the first stage(s) at baseband rates of 5.33 MHz and more.
 * out[j] = .125 * (odd + m_Xeven + 3.0 * (m_Xodd + even)) with even = x[2j], odd = x[2j + 1], m_Xeven = x[2j - 2],
 * m_Xodd = x[2j - 1]: a window of four rows, two of them delay line (rows 0, 1 of `in`), time-parallel.  The two
 * float sums first, then double arithmetic, narrowed once -- the reference's promotions (.125 and 3.0 are double
 * literals).  InLength / 2 outputs: the host refuses odd lengths (the class reads past the block then, :701). */
__global__ __launch_bounds__(256) void k_cic3(const float2* __restrict__ in, float2* __restrict__ out, unsigned n_out,
                                              unsigned C, unsigned CP, unsigned Hout)
{
  const unsigned c = blockIdx.x * 64 + threadIdx.x;
  const unsigned wy = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const unsigned o = blockIdx.y * blockDim.y + wy;
  if (c >= C || o >= n_out)
    return;
  const float2* __restrict__ p = in + (size_t)(2 * o) * CP + c;
  const float2 xe = p[0], xo = p[CP], ev = p[(size_t)2 * CP], od = p[(size_t)3 * CP];
  const float re = (float)(.125 * ((double)(od.x + xe.x) + 3.0 * (double)(xo.x + ev.x)));
  const float im = (float)(.125 * ((double)(od.y + xe.y) + 3.0 * (double)(xo.y + ev.y)));
  out[(size_t)(Hout + o) * CP + c] = make_float2(re, im);
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
#define FMD_RG 8 // (4 until round 5: 8 outputs per thread read 4.5 rows per output instead of 8 -- 0.061 against 0.092 ms alone for the RDS low-pass, +1.7 % whole path; 12 and 16: no better)
#endif
constexpr int RG = FMD_RG;
#ifndef FMD_RING_UNROLL
#define FMD_RING_UNROLL 8 // rows in flight per thread in ring_group's two long loops
#endif

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
#pragma unroll FMD_RING_UNROLL
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
#pragma unroll FMD_RING_UNROLL
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
#ifndef FMD_RP_WAVES
#define FMD_RP_WAVES 4
#endif
constexpr int RP_WAVES = FMD_RP_WAVES; // channel groups (waves) of a workgroup that share one sine table in LDS
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

__global__ __launch_bounds__(256) void k_rds_bits(const float* __restrict__ mf, unsigned R, unsigned C,
                                                 unsigned CP, RdsConsts k, ChannelState st,
                                                 uint32_t call_index, RdsGroupRec* __restrict__ queue,
                                                 unsigned* __restrict__ queue_count, unsigned queue_cap,
                                                 float* __restrict__ tap_sync, int write_taps)
{
  __builtin_amdgcn_s_setprio(3);
  const unsigned c = (blockIdx.x * blockDim.y + threadIdx.y) * 64 + threadIdx.x; // (blockDim.y groups: light_pack)
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

} // namespace fmd
