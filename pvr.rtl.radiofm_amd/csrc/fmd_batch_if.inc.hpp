/*
 * fmd_batch_if.inc.hpp -- which form of the IF stage's kernels a call launches (cFineTuner + cDownsampleFilter::
 * Process(complex), DownConvert.cpp:98-154, + RMSLevelApprox, FmDecode.cpp:505-519): outputs per workgroup, window
 * layout, load depth, tiles per workgroup and outputs per lane, by geometry and by what runs beside it.
 * Included by fmd_batch.hip in front of fmd_batch_process.inc.hpp (one translation unit).
 */
namespace
{

/* The IF FIR kernel is instantiated for a few load depths (two-sample loads in flight per lane);
 * the launch takes the smallest one that stages a tile's window in a single round trip. */
template <class IN>
using FirFn = void (*)(const typename IN::elem*, size_t, unsigned, const float2*, float2*, const float2*,
                       unsigned, unsigned, const float*, unsigned, unsigned, unsigned, unsigned, float2*,
                       unsigned, unsigned, unsigned, unsigned);

template <class IN>
using FirFn3 = void (*)(const typename IN::elem*, size_t, unsigned, const float2*, float2*, const float2*,
                        unsigned, unsigned, const float*, unsigned, unsigned, unsigned, unsigned, float2*,
                        unsigned, unsigned, unsigned, unsigned);

template <class IN, int TILE, int E, bool RB128 = false>
int launch_if_stage_t(fmd_batch* b, const void* d_iq, size_t iq_channel_stride, unsigned N, unsigned pos,
                      unsigned M, int q, hipStream_t sF, const std::function<void(int)>& mark,
                      hipEvent_t ev_start, hipEvent_t ev_stop)
{
  const fmd::Design& d = b->des;
  const unsigned C = b->C, D = d.D, T = d.table_size;
  const unsigned ntiles = (M + TILE - 1) / TILE;
  // 2^E regions of ((TILE-1)*D + order + slack) >> E slots each (see k_if_fir)
  // (+ 1: the regions of the 16-byte-read form are rounded up to an even size)
  const size_t region = ((size_t(TILE - 1) * D + d.if_order + 2u * (1u << E) + 2u) >> E) + 2u;
  // long filter: one workgroup per CU, hand-scheduled tap loop for every window layout (k_if_fir
  // LONGASM: plain window read 16 bytes at a time for D = 2 * odd, 8 bytes at a time for odd D, and
  // the two- and four-region windows)
  const bool longasm = TILE == 256 && d.if_order >= 512;
  // + 32 slots in front of the window for the tap loops' dummy last prefetch (k_if_fir WIN_PAD)
  const size_t lds = (region << E) * sizeof(float2) + (longasm ? 32 * sizeof(float2) : 0);
  if (lds > 160 * 1024)
    return fail(FMD_ERR_ARG, "IF filter window does not fit in LDS");
  // fast staging: the tuner table is a power of two that divides a tile's sample span, so a lane
  // needs the same two table entries for every load (all reference configurations: T = 64)
  const bool pow2 = (T & (T - 1)) == 0 && T <= 2u * TILE && (size_t(TILE) * D) % T == 0;
  // loads per lane needed to stage one tile in a single round trip (two samples per load)
  const unsigned rounds = unsigned(((size_t(TILE - 1) * D + d.if_order + 2) / 2 + TILE - 1) / TILE);
  FirFn<IN> kfn = &fmd::k_if_fir<IN, TILE, 1, false, E>;
  if (pow2)
    kfn = rounds <= 2 ? &fmd::k_if_fir<IN, TILE, 2, true, E>
        : rounds <= 4 ? &fmd::k_if_fir<IN, TILE, 4, true, E>
        : rounds <= 6 ? &fmd::k_if_fir<IN, TILE, 6, true, E>
        : rounds <= 7 ? &fmd::k_if_fir<IN, TILE, 7, true, E>
                      : &fmd::k_if_fir<IN, TILE, 8, true, E>;
  if (pow2 && longasm)
    kfn = &fmd::k_if_fir<IN, TILE, 8, true, E, TILE == 256 && (E == 0 || RB128), false,
                         TILE == 256 && RB128 && (E >= 1)>;
  // opt-in shuffle-reduced tap sum (not bit-exact): headline window layout only
  const bool shfl = b->params.fir_reduction == 1 && TILE == 64 && E == 0 && pow2 && rounds <= 8;
  if (shfl)
    kfn = &fmd::k_if_fir<IN, TILE, 8, true, E, false, TILE == 64 && E == 0>;
  // several tiles per workgroup with the next tile's loads in flight during the tap loop
  // (k_if_fir_mt): the headline geometry only.  Two tiles: 0.94-0.95 ms inside the pipeline against
  // 0.98-1.00 (one tile per workgroup) on the same box, the same alone; 3, 4, 8 tiles: no better
  // than one ("fir_nt" of fmd_batch_debug_set overrides, 1 = k_if_fir).
  // With the chip to itself (calls not overlapped) one tile per workgroup is the faster form (0.77
  // against 0.83 ms), and in the throughput-bound regime (> 8192 channels) the faster FIR only
  // takes from the kernels beside it (32 768 channels: 228 against 236 GS/s): two tiles only beside
  // the whole-CU serial stage.
  // (the fused multiply-add form exists for the two-tile, two-outputs-per-lane kernel only: always that one)
  const bool fma = b->params.fir_reduction == 2;
  const int fir_nt = fma ? 2 : b->dbg_fir_nt ? b->dbg_fir_nt : (b->concurrency == 2 && b->serial_exclusive ? 2 : 1);
  unsigned nblocks = C * ntiles, ntiles_l = ntiles;
  size_t lds_l = lds;
  FirFn3<IN> kfn3 = nullptr; // k_if_fir_mt3
  if (TILE == 64 && E == 0 && pow2 && rounds == 7 && fir_nt > 1 && !shfl)
  {
    const unsigned nt = fir_nt >= 8 ? 8u : fir_nt >= 4 ? 4u : fir_nt == 3 ? 3u : 2u;
    kfn = nt == 8 ? &fmd::k_if_fir_mt<IN, 7, 8>
        : nt == 4 ? &fmd::k_if_fir_mt<IN, 7, 4>
        : nt == 3 ? &fmd::k_if_fir_mt<IN, 7, 3>
                  : &fmd::k_if_fir_mt<IN, 7, 2>;
    nblocks = C * ((ntiles + nt - 1) / nt);
    // two (three) outputs per lane (k_if_fir_mt3): every sample is read from LDS once for up to two (three) taps
    const unsigned RO = fma ? 2u : unsigned(b->dbg_fir_ro), T3 = 64 * RO;
    const unsigned rounds3 = unsigned(((size_t(T3 - 1) * D + d.if_order + 2) / 2 + 63) / 64);
    if (RO > 1 && nt == 2 && d.if_order == 88 && D == 11 && (size_t(T3) * D) % T == 0 &&
        rounds3 <= (RO == 3 ? 18u : 12u))
    {
      kfn3 = RO == 3 ? &fmd::k_if_fir_mt3<IN, 18, 2, 3>
             : fma   ? &fmd::k_if_fir_mt3<IN, 12, 2, 2, 88, 11, true>
                     : &fmd::k_if_fir_mt3<IN, 12, 2, 2>;
      ntiles_l = (M + T3 - 1) / T3;
      lds_l = (size_t(T3 - 1) * D + d.if_order + 4) * sizeof(float2);
      nblocks = C * ((ntiles_l + 1) / 2);
    }
  }
  if (fma && !kfn3)
    return fail(FMD_ERR_ARG, "FMD_FIR_FMA_PARITY_WAIVED: the fused multiply-add form exists for the reference geometry "
                             "only (88-tap IF filter, downsample 11, power-of-two tuner table)");
  if (b->if_dry_run) // fmd_batch_create: only whether this geometry can be launched at all
    return FMD_OK;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const typename IN::elem* x = static_cast<const typename IN::elem*>(d_iq);
  mark(0);
  // profiled calls: the two events take the kernel's own start and stop (hipExtLaunchKernelGGL),
  // not the stream's state around it (a recorded event also counts the dispatch gap behind it)
  if (kfn3 && ev_start)
    hipExtLaunchKernelGGL(kfn3, dim3(nblocks), dim3(TILE), unsigned(lds_l), sF, ev_start, ev_stop, 0u, x,
                          iq_channel_stride, N, (const float2*)b->hist[b->hist_sel].p,
                          (float2*)b->hist[b->hist_sel ^ 1].p, (const float2*)b->lut.p, T, b->lut_idx,
                          (const float*)b->if_coeff.p, d.if_order, D, pos, M, (float2*)b->demod[q].p,
                          b->Mstride, ntiles_l, (C % 8 == 0) ? 1u : 0u, b->cpc);
  else if (kfn3)
    hipLaunchKernelGGL(kfn3, dim3(nblocks), dim3(TILE), lds_l, sF, x, iq_channel_stride, N,
                       b->hist[b->hist_sel].p, b->hist[b->hist_sel ^ 1].p, b->lut.p, T, b->lut_idx,
                       b->if_coeff.p, d.if_order, D, pos, M, b->demod[q].p, b->Mstride, ntiles_l,
                       (C % 8 == 0) ? 1u : 0u, b->cpc);
  else if (ev_start)
    hipExtLaunchKernelGGL(kfn, dim3(nblocks), dim3(TILE), unsigned(lds_l), sF, ev_start, ev_stop, 0u, x,
                          iq_channel_stride, N, (const float2*)b->hist[b->hist_sel].p,
                          (float2*)b->hist[b->hist_sel ^ 1].p, (const float2*)b->lut.p, T, b->lut_idx,
                          (const float*)b->if_coeff.p, d.if_order, D, pos, M, (float2*)b->demod[q].p,
                          b->Mstride, ntiles_l, (C % 8 == 0) ? 1u : 0u, b->cpc);
  else
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(TILE), lds_l, sF, x, iq_channel_stride, N,
                       b->hist[b->hist_sel].p, b->hist[b->hist_sel ^ 1].p, b->lut.p, T, b->lut_idx,
                       b->if_coeff.p, d.if_order, D, pos, M, b->demod[q].p, b->Mstride, ntiles_l,
                       (C % 8 == 0) ? 1u : 0u, b->cpc);
  mark(1);
  hipLaunchKernelGGL(fmd::k_if_level<IN>, dim3(C), dim3(64), 0, sF, x, iq_channel_stride, N, b->lut.p, T,
                     b->lut_idx, b->st, b->cpc);
  return FMD_OK;
}

/* Window layout by the power-of-two factor of D (k_if_fir): D odd -> plain, D = 2 * odd and
 * 4 * odd -> de-interleaved into 2 / 4 regions; higher powers of two keep 4 regions (their
 * lane stride stays even: fewer conflicts, not none). */
template <class IN, int TILE>
int launch_if_stage_e(fmd_batch* b, const void* d_iq, size_t iq_channel_stride, unsigned N, unsigned pos,
                      unsigned M, int q, hipStream_t sF, const std::function<void(int)>& mark,
                      hipEvent_t ev_start, hipEvent_t ev_stop)
{
  const unsigned D = b->des.D;
  if (D % 2 != 0)
    return launch_if_stage_t<IN, TILE, 0>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
  if (D % 4 != 0)
  { // D = 2 * odd.  Long filters: plain window read two samples at a time (fir_long_b128_asm: the b128
    // lane groups are conflict-free at this stride); otherwise the two-region window.
    const unsigned T = b->des.table_size;
    const bool pow2 = (T & (T - 1)) == 0 && T <= 2u * TILE && (size_t(TILE) * D) % T == 0;
    if (TILE == 256 && b->des.if_order >= 512 && pow2)
      return launch_if_stage_t<IN, TILE, 0>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
    return launch_if_stage_t<IN, TILE, 1>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
  }
  { // D = 4 * odd and above.  Long filters: one region fewer than the power of two in D asks for, so
    // that the lane stride inside a region stays EVEN and two adjacent positions come with one
    // 16-byte read (fir_long_e1_b128_asm / fir_long_e2_b128_asm).  Short filters: four regions (odd stride
    // for 4 * odd).
    const unsigned T = b->des.table_size;
    const bool pow2 = (T & (T - 1)) == 0 && T <= 2u * TILE && (size_t(TILE) * D) % T == 0;
    if (TILE == 256 && b->des.if_order >= 512 && pow2)
    {
      if (D % 8 != 0)
        return launch_if_stage_t<IN, TILE, 1, true>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start,
                                                    ev_stop);
      return launch_if_stage_t<IN, TILE, 2, true>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start,
                                                  ev_stop);
    }
  }
  return launch_if_stage_t<IN, TILE, 2>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
}

/* Outputs per workgroup.  Small workgroups suffer least from the serial stage: its two role waves
 * issue with priority on two SIMDs of half the CUs, a bandwidth wave sharing such a SIMD runs at a
 * fraction of its speed, and a multi-wave workgroup waits for its slowest wave.  One wave per
 * workgroup (measured, 8192 channels, in the pipeline): 0.89 ms against 0.98 ms for four waves,
 * alone 0.77 against 0.78.  Long filters keep 256 outputs per workgroup so the `order`-sample
 * halo is amortised and the window fits LDS a useful number of times. */
template <class IN>
int launch_if_stage(fmd_batch* b, const void* d_iq, size_t iq_channel_stride, unsigned N, unsigned pos,
                    unsigned M, int q, hipStream_t sF, const std::function<void(int)>& mark,
                      hipEvent_t ev_start, hipEvent_t ev_stop)
{
  const fmd::Design& d = b->des;
  const unsigned T = d.table_size;
  auto fits = [&](unsigned tile) {
    const bool pow2 = (T & (T - 1)) == 0 && T <= 2u * tile && (size_t(tile) * d.D) % T == 0;
    const size_t lds = (size_t(tile - 1) * d.D + d.if_order + 4) * sizeof(float2);
    return pow2 && d.if_order <= 4u * tile * d.D / 8u && lds <= 16 * 1024; // halo <= half the tile span
  };
  if (fits(64))
    return launch_if_stage_e<IN, 64>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
  if (fits(128))
    return launch_if_stage_e<IN, 128>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
  return launch_if_stage_e<IN, 256>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, mark, ev_start, ev_stop);
}

} // namespace
