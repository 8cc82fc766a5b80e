/*
 * fmd_k_serial.hip.h -- the baseband-rate recurrences: FM PLL, SamplesMeanRMS, pilot PLL, 38 kHz product, RDS oscillator mix
 * (k_demod_serial).
 * Part of fmd_kernels.hip.h (layout, numerics contract and citations: see there and fmd_k_common.hip.h).
 */
#pragma once

#include "fmd_k_common.hip.h"

namespace fmd
{

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

} // namespace fmd
