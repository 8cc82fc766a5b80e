/*
 * fmd_batch_process.inc.hpp -- one call of a batch: the position plan the host replays (DownConvert.cpp:112-132,
 * 203-232; FirFilter.cpp:346), the choice of kernel forms behind the IF stage, and the launches of all stages on the
 * batch's internal streams with the events that tie them (process_device_impl); the light part's launch helpers.
 * Included by fmd_batch.hip behind fmd_batch_if.inc.hpp (one translation unit: it uses that file's fmd_batch
 * struct and helpers).
 */
namespace
{

/* The kernels' constant blocks, from the design (constructor maths: FmDecode.cpp:237-314, RDSProcess.cpp:60-90). */
fmd::DemodConsts demod_consts(const fmd::Design& d)
{
  fmd::DemodConsts k{};
  k.pll_alpha = d.pll_alpha;
  k.pll_beta = d.pll_beta;
  k.nco_hl = d.nco_hl;
  k.nco_ll = d.nco_ll;
  k.demod_gain = d.demod_gain;
  k.p_minfreq = d.p_minfreq;
  k.p_maxfreq = d.p_maxfreq;
  k.p_b0 = d.p_b0;
  k.p_a1 = d.p_a1;
  k.p_a2 = d.p_a2;
  k.p_lf_b0 = d.p_lf_b0;
  k.p_lf_b1 = d.p_lf_b1;
  k.p_minsignal = d.p_minsignal;
  k.p_lock_delay = d.p_lock_delay;
  k.osc_cos = d.rds_osc_cos;
  k.osc_sin = d.rds_osc_sin;
  return k;
}

fmd::RdsConsts rds_consts(const fmd::Design& d)
{
  fmd::RdsConsts k{};
  k.pll_alpha = d.rds_pll_alpha;
  k.pll_beta = d.rds_pll_beta;
  k.nco_hl = d.rds_nco_hl;
  k.nco_ll = d.rds_nco_ll;
  k.bs_b0 = d.bitsync.b0;
  k.bs_b1 = d.bitsync.b1;
  k.bs_b2 = d.bitsync.b2;
  k.bs_a1 = d.bitsync.a1;
  k.bs_a2 = d.bitsync.a2;
  k.mf_taps = int(d.rds_mf_taps.size());
  return k;
}

fmd::AudioConsts audio_consts(const fmd::Design& d)
{
  fmd::AudioConsts k{};
  k.de_alpha = d.de_alpha;
  k.n_b0 = d.notch.b0;
  k.n_b1 = d.notch.b1;
  k.n_b2 = d.notch.b2;
  k.n_a1 = d.notch.a1;
  k.n_a2 = d.notch.a2;
  return k;
}

/* One of the post chain's two complex ring-buffer filters (cFirFilter::Process / ProcessTwo: RDS 75-tap low-pass,
 * audio 29-tap low-pass) on stream st; g0 = the ring phase at the call's first sample. */
void launch_ring2(fmd_batch* b, hipStream_t st, const float2* in, float2* out, unsigned n, unsigned T,
                  const float* taps, unsigned g0)
{
  const unsigned C = b->C, CP = b->CP;
  if (b->dbg_ring4 && T >= unsigned(fmd::RG))
    hipLaunchKernelGGL(fmd::k_ring_fir4<float2>, dim3(CP / 64, (n + 4 * fmd::RG - 1) / (4 * fmd::RG)), dim3(64, 4), 0,
                       st, in, out, n, int(T), taps, g0, C, CP, 0u, 0u);
  else
    hipLaunchKernelGGL(fmd::k_ring_fir<float2>, dim3(CP / 64, (n + fmd::RF_TI - 1) / fmd::RF_TI), dim3(64, 4),
                       size_t(T - 1 + fmd::RF_TI) * 64 * sizeof(float2), st, in, out, n, int(T), taps, g0, C, CP, 0u);
}

/* A call appends its RDS groups to queue[es]; if the queue's previous contents were handed to an
 * asynchronous drain (fmd_batch_export_rds_device), stream s first waits for that drain. */
void queue_is_free(fmd_batch* b, int es, hipStream_t s)
{
  if (b->drained_pending[es])
  {
    if (hipStreamWaitEvent(s, b->ev_drained[es], 0) != hipSuccess)
      mark_failed(b, "hipStreamWaitEvent failed in front of an RDS queue");
    b->drained_pending[es] = false;
  }
}

/* The light part of a call's post chain, lane-per-channel kernels on CP / 64 workgroups, in two halves.
 * RDS half on stream s: (layout 2: the 75-tap low-pass behind the decimator's event,) cRDSRxSignalProcessor's
 * PLL, matched filter and bit recovery; records EV_RDS. */
void launch_light_rds(fmd_batch* b, const fmd_batch::LightJob& j, hipStream_t s)
{
  const fmd::Design& d = b->des;
  const unsigned C = b->C, CP = b->CP;
  const unsigned T_mf = unsigned(d.rds_mf_taps.size());
  const dim3 rt(256);
  auto rgrid = [&](unsigned H) { return dim3((CP + 255) / 256, std::max(1u, std::min(H, 64u))); };
  if (j.lpf_here)
  { // the RDS low-pass at the head of this stream, behind the decimator
    if (hipStreamWaitEvent(s, b->cev[j.es][fmd_batch::EV_DEC], 0) != hipSuccess)
      mark_failed(b, "hipStreamWaitEvent failed in front of the RDS low-pass of a call");
    const unsigned T_lpf = unsigned(d.rds_lpf_taps.size());
    launch_ring2(b, s, b->rdsraw[j.q].p, b->rlpf[j.q].p, j.R, T_lpf, b->rds_lpf_taps.p, j.rds_lpf_g);
    hipLaunchKernelGGL(fmd::k_roll<float2>, rgrid(T_lpf - 1), rt, 0, s, b->rdsraw[j.q].p, b->rdsraw[j.q ^ 1].p,
                       T_lpf - 1, j.R, CP);
    if (hipEventRecord(b->cev[j.es][fmd_batch::EV_RDSH], s) != hipSuccess)
      mark_failed(b, "hipEventRecord failed behind the RDS low-pass of a call");
  }
  queue_is_free(b, j.es, s);
  const fmd::RdsConsts k = rds_consts(d);
  const FmdSincosTab sct{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo};
  hipLaunchKernelGGL(fmd::k_rds_pll, dim3((CP / 64 + fmd::RP_WAVES - 1) / fmd::RP_WAVES), dim3(64, fmd::RP_WAVES), 0,
                     s, b->rlpf[j.q].p, j.R, C, CP, k, b->st, b->rpll.p, T_mf - 1, b->sctab.p, sct);
  if (T_mf >= unsigned(fmd::RG))
    hipLaunchKernelGGL(fmd::k_ring_fir4<float>, dim3(CP / 64, (j.R + 4 * fmd::RG - 1) / (4 * fmd::RG)), dim3(64, 4),
                       0, s, b->rpll.p, b->rmf.p, j.R, int(T_mf), b->mf_taps2.p, j.mf_g, C, CP, 0u, 3u);
  else
    hipLaunchKernelGGL(fmd::k_ring_fir<float>, dim3(CP / 64, (j.R + fmd::RF_TI - 1) / fmd::RF_TI), dim3(64, 4),
                       size_t(T_mf - 1 + fmd::RF_TI) * 64 * sizeof(float), s, b->rpll.p, b->rmf.p, j.R, int(T_mf),
                       b->mf_taps2.p, j.mf_g, C, CP, 0u);
  hipLaunchKernelGGL(fmd::k_roll<float>, rgrid(T_mf - 1), rt, 0, s, b->rpll.p, b->rpll.p, T_mf - 1, j.R, CP);
  // Two light streams: the PREVIOUS call's status record (on the audio half's stream) copies the RDS state this
  // kernel is about to overwrite -- it goes first (its call's audio half ended a period ago: never a wait in practice)
  if (j.prev_aud && hipStreamWaitEvent(s, j.prev_aud, 0) != hipSuccess)
    mark_failed(b, "hipStreamWaitEvent failed in front of the bit recovery of a call");
  hipLaunchKernelGGL(fmd::k_rds_bits, dim3(CP / 64), dim3(64, 1), 0, s, b->rmf.p, j.R, C, CP, k, b->st,
                     j.call_index, b->queue[j.es].p, b->qcount(j.es), b->queue_cap, b->tap_sync.p, b->write_taps);
  if (j.events && hipEventRecord(b->cev[j.es][fmd_batch::EV_RDS], s) != hipSuccess)
    mark_failed(b, "hipEventRecord failed behind the RDS part of a call");
}

/* Audio half on stream s, behind event `audio_after` of the call where it has one: (layout 2: the 29-tap low-pass,)
 * de-emphasis, notch, L/R matrix, audio meter; then -- behind the RDS half -- the status record; records EV_AUD. */
void launch_light_audio(fmd_batch* b, const fmd_batch::LightJob& j, hipStream_t s)
{
  const fmd::Design& d = b->des;
  const unsigned C = b->C, CP = b->CP;
  if (j.events && j.audio_after >= 0 && hipStreamWaitEvent(s, b->cev[j.es][j.audio_after], 0) != hipSuccess)
    mark_failed(b, "hipStreamWaitEvent failed in front of the audio tail of a call");
  if (j.lpf_here)
  { // the audio low-pass in front of the tail that reads it
    const unsigned T_alp = unsigned(d.lpf_taps.size());
    launch_ring2(b, s, b->rs[j.q].p, b->alp[j.q].p, j.A, T_alp, b->audio_taps.p, j.alpf_g);
    hipLaunchKernelGGL(fmd::k_roll<float2>, dim3((CP + 255) / 256, std::max(1u, std::min(T_alp - 1, 64u))), dim3(256),
                       0, s, b->rs[j.q].p, b->rs[j.q ^ 1].p, T_alp - 1, j.A, CP);
  }
  const fmd::AudioConsts k = audio_consts(d);
  if (j.tl0) // profiling level 1: the tail's own start and stop
    hipExtLaunchKernelGGL(fmd::k_audio_tail, dim3(CP / 64), dim3(64, 1), 0u, s, j.tl0, j.tl1, 0u,
                          (const float2*)b->alp[j.q].p, j.A, C, CP, k, b->st, j.d_audio, j.audio_stride,
                          unsigned(j.sq), j.call_index);
  else
    hipLaunchKernelGGL(fmd::k_audio_tail, dim3(CP / 64), dim3(64, 1), 0, s, b->alp[j.q].p, j.A, C, CP, k, b->st,
                       j.d_audio, j.audio_stride, unsigned(j.sq), j.call_index);
  // the record's RDS state is the other half's (same stream: stream order)
  if (j.events && j.status_after_rds && hipStreamWaitEvent(s, b->cev[j.es][fmd_batch::EV_RDS], 0) != hipSuccess)
    mark_failed(b, "hipStreamWaitEvent failed in front of the status record of a call");
  hipLaunchKernelGGL(fmd::k_status_publish, dim3((C + 255) / 256), dim3(256), 0, s, b->st, C, j.call_index);
  if (j.events && hipEventRecord(b->cev[j.es][fmd_batch::EV_AUD], s) != hipSuccess)
    mark_failed(b, "hipEventRecord failed behind the audio tail of a call");
}

enum IqFormat
{
  IQ_F32 = 0, // complex<float>, the ProcessStream argument (FmDecode.h:135)
  IQ_U8 = 1   // RTL-SDR byte pairs, converted like ReadAsyncCB (RTL_SDR_Source.cpp:207-211)
};

int process_device_impl(fmd_batch* b, const void* d_iq, IqFormat fmt, size_t iq_channel_stride,
                        unsigned samples, float* d_audio, size_t audio_channel_stride,
                        unsigned* out_floats, void* stream_)
{
  if (!b || !d_iq || !d_audio)
    return fail(FMD_ERR_ARG, "fmd_batch_process_device: null argument");
  if (samples > FMD_MAX_BLOCK || samples < b->min_samples)
    return fail(FMD_ERR_SIZE, "samples must be within [fmd_batch_min_samples(), the largest block] = [" +
                                  std::to_string(b->min_samples) + ", 65536]");
  // a lane loads two IQ samples at a time: every channel's stream has to start on a pair boundary
  {
    const size_t pair = fmt == IQ_U8 ? 4 : 16;
    if ((reinterpret_cast<uintptr_t>(d_iq) % pair) || ((iq_channel_stride * (pair / 2)) % pair))
      return fail(FMD_ERR_ARG, "IQ pointer and channel stride must be multiples of two IQ samples");
  }
  const fmd::Design& d = b->des;
  const unsigned C = b->C, CP = b->CP, N = samples, D = d.D;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));
  if (int rc = check_device_errors(b)) // a failed batch takes no more calls (fmd_batch_reset clears it)
    return rc;

  /* ---- position plan (batch-uniform, mirrors the reference's bookkeeping) ---- */
  const unsigned pos = b->if_pos;
  const unsigned M = pos < N ? (N - pos + D - 1) / D : 0; // DownConvert.cpp:112,123
  if (M == 0)
    return fail(FMD_ERR_SIZE, "block shorter than the decimator phase");
  // the reference's half-band delay lines hold 32768 samples (DownConvert.cpp:267,500); longer
  // baseband blocks overrun its heap, so they are outside the contract here too
  if ((N + D - 1) / D + 51 > 32768)
    return fail(FMD_ERR_SIZE, "baseband block longer than the reference's half-band buffers (32768)");
  // Per stage of the half-band chain: how many inputs it sees and which of the reference's regimes
  // that is (see k_hb_pass / k_roll_hb_mixed): HB_PASS below L inputs, HB_MIXED below 2 (L - 1).
  enum HbMode { HB_NORMAL, HB_MIXED, HB_PASS };
  std::vector<unsigned> hb_in(d.hb.size());
  std::vector<HbMode> hb_mode(d.hb.size(), HB_NORMAL);
  unsigned R = M;
  for (size_t s = 0; s < d.hb.size(); s++)
  {
    hb_in[s] = R;
    const unsigned L = unsigned(d.hb[s].len);
    if (d.hb[s].cic)
    { // CCicN3DecimateBy2: "InLength must be an even number" (DownConvert.cpp:701) -- with an odd one the class
      // reads one sample past its block (whatever the buffer holds): outside the contract here
      if (R % 2u || R < 2u)
        return fail(FMD_ERR_SIZE, "at this baseband rate the RDS decimator starts with CIC stages: the baseband length "
                                  "of a call must be a multiple of 2 per CIC stage (the reference reads past an odd block)");
      R = R / 2; // :726
    }
    else if (L == 11)
    { // the unrolled class reads InLength - 10 .. and its first nine outputs unconditionally (:596-661)
      if (R < 20)
        return fail(FMD_ERR_SIZE, "block too short for the 11-tap half-band stage");
      R = R / 2; // :688
    }
    else if (R < L)
    {
      hb_mode[s] = HB_PASS;
      R = R / 2; // :519-520
    }
    else
    { // one output per even input index (:526-543)
      hb_mode[s] = R >= 2u * (L - 1u) ? HB_NORMAL : HB_MIXED;
      R = (R + 1) / 2;
    }
  }
  if (R == 0)
    return fail(FMD_ERR_SIZE, "block too short: no sample reaches the RDS rate");
  // fractional resampler walk (DownConvert.cpp:203-232), float arithmetic as written there
  const float p = b->rs_pos;
  const float pstep = d.rs_step;
  unsigned A = 0;
  float pf = p;
  unsigned pi = unsigned(int(pf));
  while (pi < M)
  {
    A++;
    pf = p + float(A) * pstep;
    pi = unsigned(int(pf));
  }
  float new_rs_pos = pf - float(M);
  if (new_rs_pos < 0)
    new_rs_pos = 0;
  if (A > b->Amax || M > b->Mmax)
    return fail(FMD_ERR_STATE, "internal: plan exceeds buffer geometry");
  if (A == 0)
    return fail(FMD_ERR_SIZE, "block too short: no audio frame falls into it");
  if (size_t(2) * A > audio_channel_stride && C > 1)
    return fail(FMD_ERR_ARG, "audio_channel_stride smaller than the audio produced");

  const unsigned T_lpf = unsigned(d.rds_lpf_taps.size());
  const unsigned T_mf = unsigned(d.rds_mf_taps.size());
  const unsigned T_alp = unsigned(d.lpf_taps.size());
  const unsigned Hbb = d.rs_order;
  // The call's index and everything derived from it (buffer parity, event slot) are locals until the
  // call has been submitted: a call that is refused leaves the batch exactly as it was.
  const uint32_t ci = b->call_index + 1;
  const int q = int(ci & 1u);  // buffer parity: demod, br, mix
  const int es = int(ci % fmd_batch::NSLOT); // event set / RDS queue of this call
  const int sq = int(ci & 3u); // this call's copy of the stereo flag (see ISlot)
  // call k-2 used the same buffers; its events say when they are free again
  const bool have_prev2 = ci > 2;
  hipEvent_t* pe2 = b->cev[(ci + fmd_batch::NSLOT - 2) % fmd_batch::NSLOT];
  const bool serial_mode = b->concurrency == 0 || b->profiling >= 2;
  /* "stage_mask" (fmd_batch_debug_set; results are WRONG with anything but 63): which parts of a call are
   * launched at all -- 1 IF stage, 2 serial stage, 4 half-band chain, 8 resampler, 16 / 32 the light part's RDS /
   * audio half.  Every event is still recorded, so the pipeline keeps its shape: tools/power_by_stage.py runs each
   * part alone at full rate beside a power sampler (joules per call and part). */
  const unsigned stage_mask = (!serial_mode && !b->split_post) ? unsigned(b->dbg_stage_mask) : 63u;
  hipStream_t sF = serial_mode ? stream : b->s_fir;
  hipStream_t sS = serial_mode ? stream : b->s_ser;
  hipStream_t sP = serial_mode ? stream : b->s_post;
  hipStream_t sA = sP, sR = (!serial_mode && b->split_post) ? b->s_rds : sP;
  // one-stream form: the light parts of the post chain go to their own stream (see below)
  hipStream_t sL = serial_mode ? stream : b->s_rds;
  hipEvent_t* ce = b->cev[es];
  // the first failing event operation of the call (checked once, behind the launches)
  hipError_t herr = hipSuccess;
  auto note = [&](hipError_t e) {
    if (e != hipSuccess && herr == hipSuccess)
      herr = e;
  };
  auto after = [&](hipStream_t s, hipEvent_t e) {
    if (!serial_mode)
      note(hipStreamWaitEvent(s, e, 0));
  };
  auto signal = [&](hipEvent_t e, hipStream_t s) {
    if (!serial_mode)
      note(hipEventRecord(e, s));
  };

  hipEvent_t* evset = nullptr;
  if (b->profiling && b->prof_calls < kMaxProfCalls)
  {
    const size_t need = size_t(b->prof_calls + 1) * (ST_COUNT + 1);
    while (b->ev.size() < need)
    {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      b->ev.push_back(e);
    }
    evset = &b->ev[size_t(b->prof_calls) * (ST_COUNT + 1)]; // prof_calls advances with call_index
  }
  // level 2: events between all stages (serial mode); level 1: only around the FIR kernel, on
  // the stream that kernel is launched on
  auto mark = [&](int i) { // events 0 and 1 are the FIR kernel's own start and stop (launch_if_stage)
    if (evset && b->profiling >= 2 && i > 1)
      note(hipEventRecord(evset[i], sF));
  };

  /* ---- which form the RDS decimator takes: decided here, the serial stage's form follows from it ---- */
  bool hb_all_normal = d.hb.size() <= 3;
  for (size_t s = 0; s < d.hb.size(); s++)
    hb_all_normal = hb_all_normal && hb_mode[s] == HB_NORMAL && d.hb[s].len != 11 && !d.hb[s].cic;
  /* Large batches in the usual geometries: the three half-band stages as one stream, intermediate rows in
   * LDS (k_halfband_chain).  Everything else -- short calls with a stage outside its normal regime, the
   * 11-tap class, chains of another length, small batches -- keeps one launch per stage. */
  fmd_batch::HbfPlan* hbf_pl = nullptr;
  int hbf_kind = -1; // 0: 15 / 23 / 43 taps, 1: 15 / 19 / 35
  if (hb_all_normal && d.hb.size() == 3 && b->hbf_mode != 0 && (b->hbf_mode == 1 || CP / 64 >= 64))
  {
    const int h0 = (d.hb[0].len - 1) / 2, h1 = (d.hb[1].len - 1) / 2, h2 = (d.hb[2].len - 1) / 2;
    hbf_kind = (h0 == 7 && h1 == 11 && h2 == 21) ? 0 : (h0 == 7 && h1 == 9 && h2 == 17) ? 1 : -1;
    if (hbf_kind >= 0)
    {
      const unsigned groups = CP / 64;
      const unsigned ncu = unsigned(b->n_cus) - ((b->serial_exclusive && !serial_mode) ? (groups + 1) / 2 : 0u);
      const unsigned S = std::max(1u, std::min({8u, (2u * ncu + groups / 2u) / groups, R / 32u}));
      hbf_pl = hbf_plan(b, hb_in[0], S);
    }
  }
  // ... and then the serial stage writes no mixed rows: the chain multiplies the baseband with the
  // oscillator's sequence itself (computed here, once per batch and call, and copied over on the IF stream:
  // long before anything needs it)
  const bool nomix = hbf_pl != nullptr && b->osc_on && b->dbg_nomix != 0;
  const unsigned osc_slot = ci & 3u;
  float osc_re = b->osc_re, osc_im = b->osc_im; // committed with the positions, at the end
  if (b->osc_on)
  {
    // the staging slot was last read by the copy of the call 8 calls ago: complete unless the caller has
    // submitted eight calls without ever waiting
    if (b->osc_ev_used[es])
      note(hipEventSynchronize(b->osc_ev[es]));
    float2* h = b->h_osc + size_t(es) * b->h_osc_stride;
    const float2* hp = b->h_osc + size_t((ci + fmd_batch::NSLOT - 1) % fmd_batch::NSLOT) * b->h_osc_stride;
    std::memcpy(h, hp + b->lastM, fmd_batch::kOscH * sizeof(float2)); // the previous call's last entries
    const float oc = d.rds_osc_cos, os = d.rds_osc_sin;
    float2* o = h + fmd_batch::kOscH;
    for (unsigned t = 0; t < M; t++)
    { // the statements of k_demod_serial's MIX form (this file is compiled with -ffp-contract=off too)
      const float x = osc_re * oc - osc_im * os;
      const float y = osc_im * oc + osc_re * os;
      const float gn = float(1.95 - double(osc_re * osc_re + osc_im * osc_im));
      osc_re = gn * x;
      osc_im = gn * y;
      o[t] = make_float2(x, y);
    }
    note(hipMemcpyAsync(b->osc_tab[osc_slot].p, h, (fmd_batch::kOscH + M) * sizeof(float2), hipMemcpyHostToDevice, sF));
    note(hipEventRecord(b->osc_ev[es], sF));
    b->osc_ev_used[es] = true;
  }

  /* ---- K1: tuner + IF decimating FIR  (stream F) ---- */
  /* The input is ready in the order of the caller's stream.  Where that stream has nothing pending (the usual case:
   * the host has just synchronised it for the previous outputs, or never uses it for anything else) the input IS
   * ready and no event has to carry that: one record in the caller's queue and one barrier packet in front of the IF
   * FIR less per call (+1.5 % whole path on the null stream, round 6). */
  const bool input_pending = serial_mode ? false : hipStreamQuery(stream) != hipSuccess;
  (void)hipGetLastError(); // (hipErrorNotReady is an answer, not an error)
  if (input_pending)
  {
    signal(ce[fmd_batch::EV_IN], stream);
    after(sF, ce[fmd_batch::EV_IN]);
  }
  if (have_prev2)
  {
    // (demod[q] was last read by the serial stage two calls ago: EV_SER of that call -- implied by its EV_HEAVY below,
    // the heavy part starts behind the serial stage; not where an energy experiment leaves the heavy part out)
    if (stage_mask != 63u)
      after(sF, pe2[fmd_batch::EV_SER]);
    // Also run behind the bandwidth-heavy part of the post chain of two calls ago (half-band chain, resampler):
    // side by side with those the FIR and they were both ~25 % slower.  The rest of that chain (the light part) is
    // lane-per-channel work that leaves most CUs idle: the FIR runs beside it.  (Behind the RDS half only, or
    // through a word in device memory instead of the event: measured in rounds 3 and 5, the period did not move --
    // docs/MEASUREMENTS.md.)
    after(sF, pe2[fmd_batch::EV_HEAVY]);
  }
  // a batch that is one of several sub-batches sharing these streams (fmd_batch_create above 8192 channels): the
  // FIR also stays behind the heavy part of the sub-batch call two back in the common sequence
  if (b->sched_prev2 && !serial_mode)
    after(sF, b->sched_prev2);
  {
    // EV_FIR right behind the FIR kernel (the serial stage waits for nothing else); the level meter
    // behind it also reads the input: EV_INDONE is what tells the caller its buffer is free
    const std::function<void(int)> markfn = [&](int i) {
      mark(i);
      if (i == 1)
        signal(ce[fmd_batch::EV_FIR], sF);
    };
    int rc = FMD_OK;
    if (!(stage_mask & 1u))
      markfn(1); // (energy experiment: this call leaves the IF stage out -- see "stage_mask")
    else
      rc = fmt == IQ_U8
                       ? launch_if_stage<fmd::InU8>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, markfn,
                                                   evset ? evset[0] : nullptr, evset ? evset[1] : nullptr)
                       : launch_if_stage<fmd::InF32>(b, d_iq, iq_channel_stride, N, pos, M, q, sF, markfn,
                                                    evset ? evset[0] : nullptr, evset ? evset[1] : nullptr);
    if (rc != FMD_OK) // cannot happen: the geometry was checked when the batch was created
    {
      mark_failed(b, "the IF stage refused a call after events were recorded");
      return rc;
    }
  }
  signal(ce[fmd_batch::EV_INDONE], sF);

  /* ---- K2: baseband-rate recurrences  (stream S) ---- */
  // br[q] / mix[q] were last read by the resampler / first half-band two calls ago: this call's FIR already waited
  // for that call's heavy part (EV_HEAVY above), so EV_FIR covers them.  EV_HEAVY is recorded in FRONT of the history
  // rolls behind the heavy part (the FIR starts earlier); the rolls read the tails of br[q] / mix[q]: EV_ROLL
  if (have_prev2)
    after(sS, pe2[fmd_batch::EV_ROLL]);
  after(sS, ce[fmd_batch::EV_FIR]);
  {
    const fmd::DemodConsts k = demod_consts(d);
    // From 1024 channels on the stage owns whole CUs: two channel groups per workgroup, one role wave per SIMD of a
    // CU (k_demod_serial<2, ..>; at most 64 CUs: larger batches are sub-batches); smaller batches share their CUs.
    const unsigned groups = CP / 64;
    const FmdSincosTab sct{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo};
    const unsigned Hmix = unsigned(d.hb[0].len - 1);
    /* Two groups per workgroup = one role wave on each SIMD of a CU, 64 CUs for 8192 channels.  The waves do not
     * claim their SIMD's whole register file (round 3: the bandwidth kernels' waves that fit beside the stage gain
     * more than it loses, +1-2 % whole path); "serial_claim" = 1 of fmd_batch_debug_set brings the claim back. */
    const bool serial_claim = b->dbg_serial_claim != 0;
    auto kser2 = nomix ? (serial_claim ? &fmd::k_demod_serial<2, true, false> : &fmd::k_demod_serial<2, false, false>)
                       : (serial_claim ? &fmd::k_demod_serial<2, true, true> : &fmd::k_demod_serial<2, false, true>);
    auto kser1 = nomix ? &fmd::k_demod_serial<1, false, false> : &fmd::k_demod_serial<1, false, true>;

    if (!(stage_mask & 2u))
      ; // (energy experiment: no serial stage in this call)
    else if (b->serial_exclusive && !serial_mode && evset && b->profiling == 1)
      // profiling level 1: the stage's own start and stop too (fmd_batch_debug_timeline)
      hipExtLaunchKernelGGL(kser2, dim3((groups + 1) / 2), dim3(256), 0u, sS, evset[2],
                            evset[3], 0u, (const float2*)b->demod[q].p, b->Mstride, M, C, CP, k, b->st,
                            b->brp(q), Hbb, b->mix[q].p, Hmix,
                            (const double*)(b->sctab256.p), sct, unsigned(sq),
                            (long long*)nullptr, osc_re, osc_im);
    else if (b->serial_exclusive && !serial_mode)
      hipLaunchKernelGGL(kser2, dim3((groups + 1) / 2), dim3(256), 0, sS,
                         (const float2*)b->demod[q].p, b->Mstride, M, C, CP, k, b->st, b->brp(q), Hbb, b->mix[q].p,
                         Hmix, (const double*)b->sctab256.p, sct, unsigned(sq),
                         b->serial_probe.p ? b->serial_probe.p + size_t(ci % 8) * 3 * (CP / 64) : (long long*)nullptr,
                         osc_re, osc_im);
    else
      hipLaunchKernelGGL(kser1, dim3(groups), dim3(128), 0, sS, (const float2*)b->demod[q].p,
                         b->Mstride, M, C, CP, k, b->st, b->brp(q), Hbb, b->mix[q].p, Hmix,
                         (const double*)b->sctab256.p, sct, unsigned(sq),
                         b->serial_probe.p ? b->serial_probe.p + size_t(ci % 8) * 3 * (CP / 64) : (long long*)nullptr,
                         osc_re, osc_im);
  }
  signal(ce[fmd_batch::EV_SER], sS);
  mark(2);

  const dim3 rt(256);
  auto rgrid = [&](unsigned H) { return dim3((CP + 255) / 256, std::max(1u, std::min(H, 64u))); };

  /* With overlapped calls the next call's serial stage and this call's post chain become runnable the moment this
   * call's serial stage ends.  The whole-CU serial stage needs EMPTY CUs; when the half-band kernel is dispatched first
   * it fills every CU and the serial stage starts only once those workgroups have drained (136 us late, every call).
   * So the post chain starts behind a single wave that idles for a few microseconds: the serial stage is dispatched
   * first (30-50 us after its predecessor, period 2.50 -> 2.42 ms). */
  constexpr unsigned kPostDelayUs = 20;
  // wave priority of the ring resampler in the overlapped pipeline (the half-band chain's stays 0)
  const unsigned rsr_prio = (!serial_mode && b->concurrency == 2) ? 2u : 0u;
  auto post_delay = [&](hipStream_t s) {
    if (b->serial_exclusive && !serial_mode && b->concurrency == 2)
      hipLaunchKernelGGL(fmd::k_delay, dim3(1), dim3(64), 0, s, kPostDelayUs * 100u);
  };

  /* The post chain: "heavy" = bandwidth / LDS bound and filling the chip, "light" = lane-per-channel recurrences on
   * CP / 64 workgroups.  History rolls of a chain whose stages all run in their normal regime are collected and done
   * in one launch at the chain's end (k_roll_set) instead of one launch behind every stage. */
  fmd::RollSet rolls{};
  unsigned nrolls = 0, roll_hmax = 1;
  auto roll_later = [&](const float2* src, float2* dst, unsigned H, unsigned n) {
    rolls.src[nrolls] = src;
    rolls.dst[nrolls] = dst;
    rolls.H[nrolls] = H;
    rolls.n[nrolls] = n;
    nrolls++;
    roll_hmax = std::max(roll_hmax, H);
  };
  auto roll_flush = [&](hipStream_t s) {
    if (nrolls)
      hipLaunchKernelGGL(fmd::k_roll_set, dim3((CP + 255) / 256, std::min(roll_hmax, 64u), nrolls), rt, 0, s, rolls,
                         CP);
    nrolls = 0;
    roll_hmax = 1;
  };
  /* Where the post chain's two complex low-pass filters (RDS 75 taps, audio 29 taps: ~0.05 ms each alone at 8192
   * channels, a few hundred small workgroups) and the light part run -- the stream layout of an overlapped call
   * ("lpf_late" of fmd_batch_debug_set overrides the choice):
   *   0  both filters on the heavy stream (the RDS one between half-band chain and resampler, the audio one behind
   *      the resampler), the light part on one stream behind them.  Long IF filters (> 512 taps): there the IF
   *      FIR alone sets the period and more streams only get in its way (config 5, round 6, one box: 86 500 MS/s
   *      against 80 800 with layout 2; the streams' heads wait for events in hardware queues the FIR's shares).
   *   1  the filters on a stream of their own (s_lpf), each behind the kernel that feeds it (EV_DEC: the
   *      decimator; EV_HEAVY: the resampler), the light part on one stream: round 4's layout -- the IF FIR at
   *      0.62-0.65 of the HBM peak inside the pipeline, the whole path 4 % slower than layout 2.
   *   2  no filter on the heavy stream or beside the next IF FIR's start: the RDS low-pass at the head of the light
   *      part's RDS half (s_rds, behind EV_DEC), the audio low-pass at the head of its audio half on s_lpf (behind
   *      EV_HEAVY) -- two chains of lane-per-channel kernels side by side, each shorter than a period.  Default
   *      for short IF filters: +2.3 % at the driver's flags over layout 1.
   * The filters' inputs (rdsraw, rs) are buffered by call parity, so the next call's decimator / resampler need
   * not wait for them. */
  const int layout = (serial_mode || b->split_post) ? 0
                     : b->dbg_lpf_late >= 0         ? b->dbg_lpf_late
                     : d.if_order <= 512            ? 2
                                                    : 0;
  const bool lpf_late = layout == 1, lpf_light = layout == 2;
  std::function<void()> rds_lpf_late, audio_lpf_late, mix_tail;
  hipStream_t sLPr = lpf_late ? b->s_lpf : sR;
  hipStream_t sLPa = lpf_late ? b->s_lpf : sA;
  // the resampler's form and its plan kernel (tap tables of this call's phases: no input but the positions)
  const unsigned per_step = unsigned(std::max(b->rsr_NW * b->rsr_R, 1));
  const unsigned rs_steps = (A + per_step - 1) / per_step;
  const bool ring = b->rsr_R != 0 && b->rsr_mode != 0 &&
                    (b->rsr_mode == 1 || (CP / 64 >= 64 && rs_steps >= 24 && per_step >= 16));
  bool rs_planned = false;
  auto rs_plan = [&](hipStream_t s) {
    if (!ring || rs_planned)
      return;
    rs_planned = true;
    auto go = [&](auto plan) {
      hipLaunchKernelGGL(plan, dim3(rs_steps * b->rsr_NW), dim3(64), 0, s, b->rs_coeff.p, d.rs_order, p, pstep, A,
                         b->rsr_rb, b->rsr_nbr, b->rsr_tab.p, b->rsr_nbm, b->rsr_head.p, b->rsr_steps.p);
    };
    if (b->rsr_R == 4)
      go(&fmd::k_rs_plan<4, 4>);
    else if (b->rsr_NW == 8)
      go(&fmd::k_rs_plan<2, 8>);
    else
      go(&fmd::k_rs_plan<2, 4>);
  };
  auto rds_heavy = [&]() {
    /* ---- RDS branch  (stream R): half-bands, 75-tap LPF, PLL, matched filter, bits ---- */
    after(sR, ce[fmd_batch::EV_SER]);
    // Without mixed rows stage 0 reads the last L0H history rows of br[q]: the PREVIOUS call's roll wrote
    // them, on the audio stream.  One stream (default): stream order.  Two (split_post): its EV_ROLL.
    if (nomix && sR != sA && ci > 1)
      after(sR, b->cev[(ci + fmd_batch::NSLOT - 1) % fmd_batch::NSLOT][fmd_batch::EV_ROLL]);
    post_delay(sR);
    /* Large batches in the usual geometries: the three stages as one stream, intermediate rows in LDS
     * (k_halfband_chain).  Everything else -- short calls with a stage outside its normal regime, the
     * 11-tap class, chains of another length, small batches -- keeps one launch per stage. */
    bool chain_done = false;
    if (hbf_pl)
    {
      fmd_batch::HbfPlan* pl = hbf_pl;
      const unsigned groups = CP / 64;
      const unsigned L0H = unsigned(d.hb[0].len - 1);
      const unsigned n0 = (hb_in[0] + 1) / 2, n1 = (n0 + 1) / 2;
      // stage 0's input rows and, without mixed rows, the oscillator entries that go with them: both
      // indexed by the stage's input row (0 = the first of its L0H history rows)
      const float2* in0 = nomix ? (const float2*)(b->brp(q) + size_t(Hbb - L0H) * CP) : (const float2*)b->mix[q].p;
      const float2* osc = nomix ? (const float2*)(b->osc_tab[osc_slot].p + (fmd_batch::kOscH - L0H)) : nullptr;
      auto kern = hbf_kind == 0 ? (nomix ? &fmd::k_halfband_chain<7, 11, 21, true> : &fmd::k_halfband_chain<7, 11, 21, false>)
                                : (nomix ? &fmd::k_halfband_chain<7, 9, 17, true> : &fmd::k_halfband_chain<7, 9, 17, false>);
      if (evset && b->profiling == 1 && !serial_mode) // its own start and stop (fmd_batch_debug_timeline)
        hipExtLaunchKernelGGL(kern, dim3(groups, pl->S), dim3(64, 4), 0u, sR, evset[6], evset[7], 0u, in0,
                              (const float2*)b->hbbuf[0].p, (const float2*)b->hbbuf[1].p, b->rdsraw[q].p, T_lpf - 1,
                              b->hbf_tail1.p, b->hbf_tail2.p, b->hbcoef[0], b->hbcoef[1], b->hbcoef[2],
                              (const fmd::HbStep*)pl->steps.p, (const int*)pl->seg_first.p, hb_in[0], n0, n1, C, CP,
                              osc, 0u);
      else
        hipLaunchKernelGGL(kern, dim3(groups, pl->S), dim3(64, 4), 0, sR, in0, (const float2*)b->hbbuf[0].p,
                           (const float2*)b->hbbuf[1].p, b->rdsraw[q].p, T_lpf - 1, b->hbf_tail1.p, b->hbf_tail2.p,
                           b->hbcoef[0], b->hbcoef[1], b->hbcoef[2], (const fmd::HbStep*)pl->steps.p,
                           (const int*)pl->seg_first.p, hb_in[0], n0, n1, C, CP, osc, 0u);
      if (nomix) // the next call's stage-0 history, should it take a launch per stage (it reads mixed rows)
        mix_tail = [&, L0H]() {
          hipLaunchKernelGGL(fmd::k_mix_tail, rgrid(L0H), rt, 0, sR,
                             (const float2*)(b->brp(q) + size_t(Hbb + hb_in[0] - L0H) * CP),
                             (const float2*)(b->osc_tab[osc_slot].p + fmd_batch::kOscH + hb_in[0] - L0H),
                             b->mix[q ^ 1].p, L0H, CP);
        };
      else
        roll_later(b->mix[q].p, b->mix[q ^ 1].p, L0H, hb_in[0]);
      roll_later(b->hbf_tail1.p, b->hbbuf[0].p, unsigned(d.hb[1].len - 1), 0u);
      roll_later(b->hbf_tail2.p, b->hbbuf[1].p, unsigned(d.hb[2].len - 1), 0u);
      chain_done = true;
    }
    if (!chain_done)
    {
      const float2* in = b->mix[q].p;
      for (size_t s = 0; s < d.hb.size(); s++)
      {
        const unsigned n_out =
            (d.hb[s].cic || d.hb[s].len == 11 || hb_mode[s] == HB_PASS) ? hb_in[s] / 2 : (hb_in[s] + 1) / 2;
        const bool last = (s + 1 == d.hb.size());
        float2* outp = last ? b->rdsraw[q].p : b->hbbuf[s].p;
        const unsigned Hout = last ? (T_lpf - 1) : unsigned(d.hb[s + 1].len - 1);
        const int hb4 = b->dbg_hb4;
        const unsigned Hs = unsigned(d.hb[s].len - 1);
        float2* const hist_dst = s == 0 ? b->mix[q ^ 1].p : b->hbbuf[s - 1].p; // where the delay line lives
        if (hb_mode[s] == HB_PASS)
        { // unfiltered; the delay line stays (stage 0 keeps it in the other parity's buffer: copy it over)
          hipLaunchKernelGGL(fmd::k_hb_pass, rgrid(n_out), rt, 0, sR, in, Hs, outp, Hout, n_out, CP);
          if (s == 0)
            hipLaunchKernelGGL(fmd::k_roll<float2>, rgrid(Hs), rt, 0, sR, b->mix[q].p, b->mix[q ^ 1].p, Hs, 0u, CP);
          in = outp;
          continue;
        }
        if (d.hb[s].cic)
          hipLaunchKernelGGL(fmd::k_cic3, dim3(CP / 64, (n_out + 3) / 4), dim3(64, 4), 0, sR, in, outp, n_out, C, CP,
                             Hout);
        else if (d.hb[s].len == 11)
          hipLaunchKernelGGL(fmd::k_halfband11, dim3(CP / 64, (n_out + 3) / 4), dim3(64, 4), 0, sR, in, outp,
                             n_out, b->hbcoef[s], C, CP, Hout);
        else if (hb4 && (d.hb[s].len - 1) / 2 >= 4 && d.hb[s].len <= 55)
          hipLaunchKernelGGL(fmd::k_halfband4, dim3(CP / 64, (n_out + 15) / 16), dim3(64, 4), 0, sR, in, outp,
                             n_out, d.hb[s].len, b->hbcoef[s], C, CP, Hout);
        else
        hipLaunchKernelGGL(fmd::k_halfband, dim3(CP / 64, (n_out + 4 * fmd::HB_R - 1) / (4 * fmd::HB_R)),
                           dim3(64, 4), 0, sR, in, outp, n_out, d.hb[s].len, b->hbcoef[s], C, CP, Hout);
        // keep the last L-1 input rows of this stage for the next call, then its input is free
        if (hb_mode[s] == HB_MIXED)
          hipLaunchKernelGGL(fmd::k_roll_hb_mixed, dim3((CP + 255) / 256), rt, 0, sR, in, (const float2*)outp,
                             hist_dst, Hs, hb_in[s], n_out, Hout, CP);
        else if (s == 0)
        { // tail of mix[q] -> history rows of mix[q^1], which the next call's half-band reads
          if (hb_all_normal)
            roll_later(b->mix[q].p, b->mix[q ^ 1].p, Hs, hb_in[0]);
          else
            hipLaunchKernelGGL(fmd::k_roll<float2>, rgrid(Hs), rt, 0, sR, b->mix[q].p, b->mix[q ^ 1].p, Hs,
                               hb_in[0], CP);
        }
        else if (hb_all_normal)
          roll_later(b->hbbuf[s - 1].p, b->hbbuf[s - 1].p, Hs, hb_in[s]);
        else
          hipLaunchKernelGGL(fmd::k_roll<float2>, rgrid(Hs), rt, 0, sR, b->hbbuf[s - 1].p,
                             b->hbbuf[s - 1].p, Hs, hb_in[s], CP);
        in = outp;
      }
    }
    mark(3);
    if (lpf_light) // the low-pass is the light part's (launch_light_rds); the decimator's rolls wait for the
    {              // resampler's: one launch behind EV_HEAVY (k_roll_set takes four)
      if (nrolls > 3)
        roll_flush(sR);
      return;
    }
    auto lpf = [&]() {
      launch_ring2(b, sLPr, b->rdsraw[q].p, b->rlpf[q].p, R, T_lpf, b->rds_lpf_taps.p, b->rds_lpf_g);
      if (hb_all_normal)
      {
        roll_later(b->rdsraw[q].p, b->rdsraw[q ^ 1].p, T_lpf - 1, R);
        roll_flush(sLPr);
      }
      else
        hipLaunchKernelGGL(fmd::k_roll<float2>, rgrid(T_lpf - 1), rt, 0, sLPr, b->rdsraw[q].p, b->rdsraw[q ^ 1].p,
                           T_lpf - 1, R, CP);
    };
    if (lpf_late)
    { // the low-pass later, on its own stream (see above)
      if (nrolls > 3)
        roll_flush(sR);
      rds_lpf_late = lpf;
      return;
    }
    if (mix_tail)
    {
      mix_tail();
      mix_tail = nullptr;
    }
    lpf();
    mark(4);
  };
  auto audio_heavy = [&]() {

    /* ---- audio branch  (stream A): resamplers, 15 kHz LPF, de-emphasis / notch / matrix ---- */
    if (sA != sR) // (one stream: the RDS branch in front has waited)
      after(sA, ce[fmd_batch::EV_SER]);
    /* Large batches stream the rows through an LDS ring (k_resample_ring: every row crosses the fabric
     * once per segment instead of ~6 times); small ones, short calls and geometries whose window does
     * not fit a CU's LDS keep the window-per-wave form, which has more workgroups to offer. */
    if (ring)
    {
      // one workgroup (a whole CU's LDS) for every CU the serial stage leaves free: one round, with an equal
      // run of (group, step) units each, >= 8 steps (measured with 160 / 176 / 184 / 192 of 192: 267 800 /
      // 278 700 / 280 000 / 282 300 MS/s whole path on one box)
      const unsigned groups = CP / 64;
      const unsigned ncu = unsigned(b->n_cus) - ((b->serial_exclusive && !serial_mode) ? (groups + 1) / 2 : 0u);
      const unsigned units = groups * rs_steps;
      unsigned W = std::max(1u, ncu);
      W = std::max(1u, std::min(W, units / 8u));
      const unsigned per_wg = (units + W - 1) / W;
      W = (units + per_wg - 1) / per_wg;
      const unsigned lds = b->rsr_nbr * 4096u;
      rs_plan(sA); // (already done in front of the half-band chain where the two share a stream)
      auto go = [&](auto kern) {
        if (evset && b->profiling == 1 && !serial_mode)
          hipExtLaunchKernelGGL(kern, dim3(W), dim3(64, b->rsr_NW + 1), lds, sA, evset[8], evset[9], 0u,
                                (const float2*)b->brp(q), Hbb, b->rsr_rb, d.rs_order, (const float*)b->rsr_tab.p,
                                b->rsr_nbm, (const int*)b->rsr_head.p, (const int*)b->rsr_steps.p, rs_steps, per_wg,
                                b->rsr_nbr, A, b->rs[q].p, T_alp - 1, C, CP, rsr_prio);
        else
        hipLaunchKernelGGL(kern, dim3(W), dim3(64, b->rsr_NW + 1), lds, sA, b->brp(q), Hbb, b->rsr_rb,
                           d.rs_order, b->rsr_tab.p, b->rsr_nbm, b->rsr_head.p, b->rsr_steps.p, rs_steps, per_wg,
                           b->rsr_nbr, A, b->rs[q].p, T_alp - 1, C, CP, rsr_prio);
      };
      if (b->rsr_R == 4)
        go(&fmd::k_resample_ring<4, 4>);
      else if (b->rsr_NW == 8 && b->params.fir_reduction == 2)
        go(&fmd::k_resample_ring<2, 8, true>); // FMD_FIR_FMA_PARITY_WAIVED
      else if (b->rsr_NW == 8)
        go(&fmd::k_resample_ring<2, 8>);
      else
        go(&fmd::k_resample_ring<2, 4>);
    }
    else
    {
      hipLaunchKernelGGL(fmd::k_rs_table, dim3(A), dim3(64), 0, sA, b->rs_coeff.p, d.rs_order, p,
                         pstep, A, b->ktab.p, b->rs_row, b->rs_margin, b->pidx.p);
      hipLaunchKernelGGL(fmd::k_resample, dim3(CP / 64, (A + 4 * fmd::RS_R - 1) / (4 * fmd::RS_R)),
                         dim3(64, 4), 0, sA, b->brp(q), Hbb, d.rs_order, b->ktab.p, b->rs_row,
                         b->rs_margin, b->pidx.p, A, b->rs[q].p, T_alp - 1, C, CP);
    }
    roll_later(b->brp(q), b->brp(q ^ 1), Hbb, M); // with the low-pass's own roll, at the chain's end
    mark(6);
    if (lpf_light) // the low-pass is the light part's (launch_light_audio); the baseband rows' roll: below
      return;
    auto lpf = [&]() {
      launch_ring2(b, sLPa, b->rs[q].p, b->alp[q].p, A, T_alp, b->audio_taps.p, b->alpf_g);
      roll_later(b->rs[q].p, b->rs[q ^ 1].p, T_alp - 1, A);
      roll_flush(sLPa);
    };
    if (lpf_late)
    { // the low-pass (and its own roll) later; the baseband rows' history behind EV_HEAVY (below)
      audio_lpf_late = lpf;
      return;
    }
    lpf();
    mark(7);
  };
  // the light part's two halves (launch_light_rds / launch_light_audio), from the call's values
  fmd_batch::LightJob job;
  job.events = !serial_mode;
  job.lpf_here = lpf_light;
  job.rds_lpf_g = b->rds_lpf_g;
  job.alpf_g = b->alpf_g;
  job.R = R;
  job.A = A;
  job.mf_g = b->mf_g;
  job.q = q;
  job.es = es;
  job.sq = sq;
  job.call_index = ci;
  job.d_audio = d_audio;
  job.audio_stride = audio_channel_stride;
  if (serial_mode || b->split_post)
  { // stage order of the reference (what the per-stage profile is keyed to), or two streams
    rds_heavy();
    if (sA != sR && !serial_mode && ci > 1) // call k-1's status record copies the RDS state this call's bit recovery writes
      job.prev_aud = b->cev[(ci + fmd_batch::NSLOT - 1) % fmd_batch::NSLOT][fmd_batch::EV_AUD];
    launch_light_rds(b, job, sR); // (records EV_RDS)
    mark(5);
    audio_heavy();
    job.audio_after = -1; // behind its resampler and low-pass in stream order
    job.status_after_rds = sA != sR;
    launch_light_audio(b, job, sA); // (records EV_AUD)
    mark(8);
    signal(ce[fmd_batch::EV_HEAVY], sA);
    signal(ce[fmd_batch::EV_ROLL], sA);
  }
  else
  { // Both heavy parts first on the post stream, the light parts behind them on streams of their
    // own: the next call's FIR runs beside the light parts, and the next call's heavy parts do
    // not queue behind them (rlpf / alp, the buffers between a heavy and a light part, are
    // double-buffered by call parity; their readers of two calls ago are long done).
    if (have_prev2)
    {
      after(sP, pe2[fmd_batch::EV_RDS]);
      after(sA, pe2[fmd_batch::EV_AUD]);
    }
    if (lpf_late && have_prev2)
    { // rdsraw[q] / rs[q] are written again: their low-pass filters of two calls ago have read them
      after(sP, pe2[fmd_batch::EV_RDSH]);
      after(sP, pe2[fmd_batch::EV_ALP]);
    }
    if (lpf_light && have_prev2) // (rs[q]'s reader of two calls ago is in front of EV_AUD, waited for above)
      after(sP, pe2[fmd_batch::EV_RDSH]);
    if (lpf_late || lpf_light)
      rs_plan(sP); // off the path between the half-band chain and the resampler
    if (stage_mask & 4u)
      rds_heavy();
    signal(ce[(lpf_late || lpf_light) ? fmd_batch::EV_DEC : fmd_batch::EV_RDSH], sR);
    if (stage_mask & 8u)
      audio_heavy();
    signal(ce[fmd_batch::EV_HEAVY], sP); // the next-but-one call's IF FIR may go
    if (mix_tail)
    {
      mix_tail();
      mix_tail = nullptr;
    }
    roll_flush(sP); // (layouts 1, 2: the history rolls of both heavy parts in one launch)
    signal(ce[fmd_batch::EV_ROLL], sP);
    if (lpf_late)
    { // the two low-pass filters on their own stream
      hipStream_t sl = b->s_lpf;
      if (have_prev2)
      { // rlpf[q] / alp[q] were last read by the light part of two calls ago
        after(sl, pe2[fmd_batch::EV_RDS]);
        after(sl, pe2[fmd_batch::EV_AUD]);
      }
      after(sl, ce[fmd_batch::EV_DEC]);
      rds_lpf_late();
      signal(ce[fmd_batch::EV_RDSH], sl);
      after(sl, ce[fmd_batch::EV_HEAVY]);
      audio_lpf_late();
      signal(ce[fmd_batch::EV_ALP], sl);
    }
    if (evset && b->profiling == 1)
    {
      job.tl0 = evset[4];
      job.tl1 = evset[5];
    }
    /* The light part goes out at once: its RDS half behind the RDS half of the heavy part (it runs beside the
     * resampler), the audio tail behind the whole heavy part.  (Until round 3 it was kept back until the NEXT
     * call's serial stage had ended, so that it ran beside that call's heavy part and not beside a FIR -- built
     * again and measured in round 5: the FIR gains 0.02 of the HBM peak, the whole path loses 7.5 %; removed.) */
    // layout 2: the audio half on the stream the filters have in layout 1, beside the RDS half
    hipStream_t s_aud = lpf_light ? b->s_lpf : sL;
    if (!lpf_light)
      after(sL, ce[fmd_batch::EV_RDSH]);
    if (stage_mask & 16u) // (stage_mask: the energy experiment's half-by-half runs)
    {
      if (s_aud != sL && ci > 1) // call k-1's status record copies the RDS state this call's bit recovery writes
        job.prev_aud = b->cev[(ci + fmd_batch::NSLOT - 1) % fmd_batch::NSLOT][fmd_batch::EV_AUD];
      launch_light_rds(b, job, sL);
    }
    else
      signal(ce[fmd_batch::EV_RDS], sL);
    job.audio_after = lpf_late ? fmd_batch::EV_ALP : fmd_batch::EV_HEAVY;
    job.status_after_rds = s_aud != sL;
    if (stage_mask & 32u)
      launch_light_audio(b, job, s_aud);
    else
      signal(ce[fmd_batch::EV_AUD], s_aud);
  }
  mark(9);
  if (!serial_mode && b->concurrency < 2)
  { // order the caller's stream after everything this call launched
    note(hipStreamWaitEvent(stream, ce[fmd_batch::EV_AUD], 0));
    note(hipStreamWaitEvent(stream, ce[fmd_batch::EV_RDS], 0));
    note(hipStreamWaitEvent(stream, ce[fmd_batch::EV_INDONE], 0));
    note(hipStreamWaitEvent(stream, ce[fmd_batch::EV_ROLL], 0));
  }
  note(hipGetLastError());
  if (herr != hipSuccess)
  { // part of the call is on the device, part is not: histories and channel state no longer line up
    b->failed = true;
    b->fail_msg = std::string("a call broke off while it was being submitted: ") + hipGetErrorString(herr);
    return fail(FMD_ERR_DEVICE, b->fail_msg);
  }
  /* ---- the call is submitted: commit its index together with the positions ---- */
  b->call_index = ci;
  b->slot_call[es] = ci;
  if (evset)
    b->prof_calls++;

  /* ---- advance the host-tracked positions ---- */
  b->if_pos = pos + M * D - N;                      // DownConvert.cpp:132
  b->lut_idx = (b->lut_idx + N) % d.table_size;     // FmDecode.cpp:81
  b->rs_pos = new_rs_pos;                           // DownConvert.cpp:230-232
  b->osc_re = osc_re;                               // DownConvert.cpp:440-441 (batch-wide, see osc_on)
  b->osc_im = osc_im;
  b->rds_lpf_g = (b->rds_lpf_g + R) % T_lpf;
  b->mf_g = (b->mf_g + R) % T_mf;
  b->alpf_g = (b->alpf_g + A) % T_alp;
  b->hist_sel ^= 1;
  b->lastM = M;
  b->lastA = A;
  b->lastR = R;
  if (out_floats)
    *out_floats = 2 * A;
  return FMD_OK;
}

} // namespace
