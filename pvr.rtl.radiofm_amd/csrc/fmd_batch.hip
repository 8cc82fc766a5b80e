/*
 * fmd_batch.hip -- C ABI (include/fmd.h) of the MI355X FM decoder: device memory, the
 * per-call position plan (host) and the kernel sequence.  gfx950 only, no CPU fallback.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared fmd_batch.hip -o libfmd_hip.so
 */
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fmd_design.hpp"
#include "fmd_groups.hpp"
#include "fmd_kernels.hip.h"

namespace
{

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                            \
  do                                                                                            \
  {                                                                                             \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(FMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

enum Stage
{
  ST_IF_FIR = 0,
  ST_DEMOD_SERIAL,
  ST_RDS_HALFBAND,
  ST_RDS_LPF,
  ST_RDS_SERIAL,
  ST_RESAMPLE,
  ST_AUDIO_LPF,
  ST_AUDIO_TAIL,
  ST_ROLL,
  ST_COUNT
};
const char* kStageNames[ST_COUNT] = {"if_fir",      "demod_serial", "rds_halfband",
                                     "rds_lpf",     "rds_serial",   "resample",
                                     "audio_lpf",   "audio_tail",   "history_roll"};

template <typename T>
struct DevBuf
{
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count)
  {
    n = count;
    if (hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess)
      return -1;
    return hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) == hipSuccess ? 0 : -1;
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
  }
};

} // namespace

struct fmd_batch
{
  fmd::Params params;
  fmd::Design des;
  unsigned C = 0, CP = 0;
  int device = 0;
  fmd_callbacks cb{};
  void* user = nullptr;
  std::vector<int> shifts;

  // host-tracked, batch-uniform positions
  unsigned if_pos = 0;     // cDownsampleFilter::m_pos_int
  unsigned lut_idx = 0;    // cFineTuner::m_index
  float rs_pos = 0.0f;     // cDownsampleFilter::m_pos_frac (mono == stereo)
  unsigned rds_lpf_g = 0;  // samples since init of the RDS LPF, mod taps
  unsigned mf_g = 0;       // samples since init of the RDS matched filter, mod taps
  unsigned alpf_g = 0;     // samples since init of the audio LPF, mod taps
  int hist_sel = 0;        // IF history ping-pong
  uint32_t call_index = 0;

  // geometry
  unsigned Mmax = 0, Mstride = 0, Amax = 0, Rmax = 0;
  std::vector<unsigned> hb_nmax; // max input length per HB stage
  // last call
  unsigned lastM = 0, lastA = 0, lastR = 0;

  // device memory
  DevBuf<float2> lut, hist[2], demod, br, mix, rdsraw, rlpf, rs, alp;
  std::vector<DevBuf<float2>> hbbuf; // input buffers of stages 1..n-1 (stage 0 reads mix)
  DevBuf<float> if_coeff, rs_coeff, rds_lpf_taps, mf_taps2, audio_taps, ktab;
  DevBuf<float> rpll, rmf, tap_sync;
  DevBuf<double> sctab;
  DevBuf<int> pidx;
  DevBuf<float> fstate; // all float state arrays, CP each
  DevBuf<int> istate;
  DevBuf<uint16_t> r_data;
  DevBuf<fmd::RdsGroupRec> queue;
  DevBuf<unsigned> queue_count;
  unsigned queue_cap = 0;
  fmd::ChannelState st{};
  std::vector<fmd::HbCoef> hbcoef;

  // host staging for the host-buffer entry point
  DevBuf<float> h_iq, h_audio;
  size_t h_iq_cap = 0, h_audio_cap = 0;

  std::vector<std::unique_ptr<fmd::GroupDecoder>> gdec;

  // profiling: 0 off, 1 = events around the IF FIR kernel only, 2 = around every stage.
  // One event set per call (up to kMaxProfCalls) so nothing has to synchronise inside a timed loop.
  int profiling = 0;
  int write_taps = 0; // stage taps of the RDS recurrences are only written on request
  std::vector<hipEvent_t> ev; // [calls][ST_COUNT + 1]
  unsigned prof_calls = 0;

  ~fmd_batch()
  {
    (void)hipSetDevice(device);
    lut.release();
    hist[0].release();
    hist[1].release();
    demod.release();
    mix.release();
    rdsraw.release();
    rlpf.release();
    rs.release();
    alp.release();
    for (auto& b : hbbuf)
      b.release();
    if_coeff.release();
    rs_coeff.release();
    br.release();
    rds_lpf_taps.release();
    mf_taps2.release();
    audio_taps.release();
    ktab.release();
    rpll.release();
    rmf.release();
    tap_sync.release();
    sctab.release();
    pidx.release();
    fstate.release();
    istate.release();
    r_data.release();
    queue.release();
    queue_count.release();
    h_iq.release();
    h_audio.release();
    for (auto& e : ev)
      (void)hipEventDestroy(e);
  }
};

namespace
{

constexpr unsigned kMaxProfCalls = 512;

int upload(void* dst, const void* src, size_t bytes)
{
  return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

void bind_state(fmd_batch* b)
{
  b->st.f = b->fstate.p;
  b->st.i = b->istate.p;
  b->st.r_data = b->r_data.p;
  b->st.CP = b->CP;
}

/* state a freshly constructed cFmDecoder has (ctor values that are not zero) */
int init_signal_state(fmd_batch* b)
{
  const size_t CP = b->CP;
  std::vector<float> v(CP);
  std::fill(v.begin(), v.end(), b->des.p_freq0); // cPilotPhaseLock: m_freq = freq*2pi (:131)
  if (upload(b->st.F(fmd::F_P_FREQ), v.data(), CP * sizeof(float)))
    return -1;
  std::fill(v.begin(), v.end(), 1.0f); // CRDSDownConvert: m_Osc1 = (1, 0) (DownConvert.cpp:284)
  if (upload(b->st.F(fmd::F_OSC_RE), v.data(), CP * sizeof(float)))
    return -1;
  return 0;
}

template <typename T>
int zero_rows(T* p, size_t rows, size_t CP)
{
  return hipMemset(p, 0, rows * CP * sizeof(T)) == hipSuccess ? 0 : -1;
}

/* cFmDecoder::Reset (FmDecode.cpp:326-338) + cRDSRxSignalProcessor::Reset (RDSProcess.cpp:92-118):
 * clears the demod/RDS recurrences and re-initialises the three RDS filters; leaves tuner index,
 * FIR/resampler histories, pilot PLL, half-band histories, oscillator, de-emphasis, notch,
 * audio LPF and the block-sync shift register untouched, like the reference. */
int do_reset(fmd_batch* b)
{
  const size_t CP = b->CP;
  using namespace fmd;
  const ChannelState& s = b->st;
  const int fz[] = {F_IF_LEVEL, F_BB_MEAN, F_BB_LEVEL, F_DC_OFF, F_NCO_INCR, F_NCO_PHASE, F_R_PHASE,
                    F_R_FREQ, F_R_W1, F_R_W2, F_R_LAST_SYNC, F_R_LAST_SLOPE, F_R_LAST_DATA};
  for (int slot : fz)
    if (hipMemset(s.F(slot), 0, CP * sizeof(float)) != hipSuccess)
      return -1;
  const int iz[] = {I_STEREO, I_R_LAST_BIT, I_R_BITPOS, I_R_BLOCK, I_R_STATE, I_R_BOFF};
  for (int slot : iz)
    if (hipMemset(s.I(slot), 0, CP * sizeof(int)) != hipSuccess)
      return -1;
  // RDS LPF ring (history rows of rdsraw), matched filter ring, positions
  if (zero_rows(b->rdsraw.p, b->des.rds_lpf_taps.size() - 1, CP))
    return -1;
  if (zero_rows(b->rpll.p, b->des.rds_mf_taps.size() - 1, CP))
    return -1;
  b->rds_lpf_g = 0;
  b->mf_g = 0;
  for (auto& g : b->gdec)
    if (g)
      g->reset();
  return 0;
}

} // namespace

extern "C" {

const char* fmd_last_error(void)
{
  return g_err.c_str();
}

const char* fmd_version(void)
{
  return "fmd-hip 0.1 (gfx950)";
}

const char* fmd_stage_name(unsigned idx)
{
  return idx < ST_COUNT ? kStageNames[idx] : "";
}

int fmd_batch_create(const fmd_params* params, unsigned n_channels, const int* tuning_shifts,
                     int device, const fmd_callbacks* cb, void* user, fmd_batch** out)
{
  if (!params || !out || n_channels == 0)
    return fail(FMD_ERR_ARG, "fmd_batch_create: null argument or zero channels");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FMD_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev)
    return fail(FMD_ERR_ARG, "fmd_batch_create: device ordinal out of range");
  HIPCHK(hipSetDevice(device));

  std::unique_ptr<fmd_batch> b(new fmd_batch);
  b->device = device;
  b->params.sample_rate_if = params->sample_rate_if;
  b->params.tuning_offset = params->tuning_offset;
  b->params.sample_rate_pcm = params->sample_rate_pcm;
  b->params.bandwidth_pcm = params->bandwidth_pcm;
  b->params.downsample = params->downsample;
  b->params.us_version = params->us_version != 0;
  b->params.table_size = params->table_size;
  b->params.if_filter_order = params->if_filter_order;
  try
  {
    b->des = fmd::make_design(b->params);
  }
  catch (const std::exception& e)
  {
    return fail(FMD_ERR_ARG, e.what());
  }
  const fmd::Design& d = b->des;
  if (d.if_order > FMD_MIN_BLOCK)
    return fail(FMD_ERR_ARG, "if_filter_order larger than the minimum block");
  for (const auto& h : d.hb)
    if (h.len == 11)
      return fail(FMD_ERR_ARG, "baseband rate >= 320 kHz needs the 11-tap half-band class (unsupported)");
  if (cb)
    b->cb = *cb;
  b->user = user;
  const unsigned C = n_channels;
  const unsigned CP = (C + 63u) & ~63u;
  b->C = C;
  b->CP = CP;

  // cFineTuner tables, one per channel
  b->shifts.resize(C);
  const int def_shift = fmd::tuning_shift_for(b->params);
  std::vector<float> lut(size_t(2) * d.table_size * C);
  for (unsigned c = 0; c < C; c++)
  {
    b->shifts[c] = tuning_shifts ? tuning_shifts[c] : def_shift;
    if (c > 0 && b->shifts[c] == b->shifts[c - 1])
      std::copy_n(&lut[size_t(2) * d.table_size * (c - 1)], 2 * d.table_size, &lut[size_t(2) * d.table_size * c]);
    else
    {
      auto t = fmd::make_tuner_lut(d.table_size, b->shifts[c]);
      std::copy(t.begin(), t.end(), &lut[size_t(2) * d.table_size * c]);
    }
  }

  // geometry
  b->Mmax = (FMD_MAX_BLOCK + d.D - 1) / d.D + 1;
  b->Mstride = (b->Mmax + 15u) & ~15u;
  b->Amax = unsigned(double(b->Mmax) / double(d.rs_step)) + 4;
  {
    unsigned n = b->Mmax;
    for (size_t s = 0; s < d.hb.size(); s++)
    {
      b->hb_nmax.push_back(n);
      n = (n + 1) / 2;
    }
    b->Rmax = n;
  }
  const unsigned T_lpf = unsigned(d.rds_lpf_taps.size());
  const unsigned T_mf = unsigned(d.rds_mf_taps.size());
  const unsigned T_alp = unsigned(d.lpf_taps.size());

  int bad = 0;
  bad |= b->lut.alloc(size_t(d.table_size) * C);
  bad |= b->hist[0].alloc(size_t(d.if_order) * C);
  bad |= b->hist[1].alloc(size_t(d.if_order) * C);
  bad |= b->demod.alloc(size_t(b->Mstride) * C);
  bad |= b->if_coeff.alloc(d.if_coeff.size());
  bad |= b->rs_coeff.alloc(d.rs_coeff.size());
  bad |= b->br.alloc(size_t(d.rs_order + b->Mmax) * CP);
  if (d.hb.empty())
    return fail(FMD_ERR_ARG, "baseband rate too low for the RDS decimation chain");
  bad |= b->mix.alloc(size_t(d.hb[0].len - 1 + b->Mmax) * CP);
  b->hbbuf.resize(d.hb.size() - 1);
  for (size_t s = 1; s < d.hb.size(); s++)
    bad |= b->hbbuf[s - 1].alloc(size_t(d.hb[s].len - 1 + b->hb_nmax[s]) * CP);
  bad |= b->rdsraw.alloc(size_t(T_lpf - 1 + b->Rmax) * CP);
  bad |= b->rlpf.alloc(size_t(b->Rmax) * CP);
  bad |= b->rpll.alloc(size_t(T_mf - 1 + b->Rmax) * CP);
  bad |= b->rmf.alloc(size_t(b->Rmax) * CP);
  bad |= b->tap_sync.alloc(size_t(b->Rmax) * CP);
  bad |= b->rs.alloc(size_t(T_alp - 1 + b->Amax) * CP);
  bad |= b->alp.alloc(size_t(b->Amax) * CP);
  bad |= b->rds_lpf_taps.alloc(T_lpf);
  bad |= b->mf_taps2.alloc(size_t(2) * T_mf);
  bad |= b->audio_taps.alloc(T_alp);
  bad |= b->ktab.alloc(size_t(b->Amax) * (d.rs_order + 1));
  bad |= b->pidx.alloc(b->Amax);
  bad |= b->sctab.alloc(d.sincos_tab.size());
  bad |= b->fstate.alloc(size_t(fmd::F_SLOTS) * CP);
  bad |= b->istate.alloc(size_t(fmd::I_SLOTS) * CP);
  bad |= b->r_data.alloc(size_t(4) * CP);
  b->queue_cap = std::max(4096u, 8u * C);
  bad |= b->queue.alloc(b->queue_cap);
  bad |= b->queue_count.alloc(1);
  if (bad)
    return fail(FMD_ERR_DEVICE, std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError()));

  bad |= upload(b->lut.p, lut.data(), lut.size() * sizeof(float));
  bad |= upload(b->if_coeff.p, d.if_coeff.data(), d.if_coeff.size() * sizeof(float));
  bad |= upload(b->rs_coeff.p, d.rs_coeff.data(), d.rs_coeff.size() * sizeof(float));
  bad |= upload(b->rds_lpf_taps.p, d.rds_lpf_taps.data(), T_lpf * sizeof(float));
  bad |= upload(b->audio_taps.p, d.lpf_taps.data(), T_alp * sizeof(float));
  bad |= upload(b->sctab.p, d.sincos_tab.data(), d.sincos_tab.size() * sizeof(double));
  bad |= upload(b->mf_taps2.p, d.rds_mf_taps.data(), T_mf * sizeof(float));
  for (const auto& h : d.hb)
  {
    fmd::HbCoef hc{};
    for (int i = 0; i < h.len; i++)
      hc.c[i] = h.coef[size_t(i)];
    b->hbcoef.push_back(hc);
  }
  bind_state(b.get());
  bad |= init_signal_state(b.get());
  if (bad)
    return fail(FMD_ERR_DEVICE, "upload of constants failed");
  if (do_reset(b.get())) // the cFmDecoder ctor ends with Reset() (FmDecode.cpp:313)
    return fail(FMD_ERR_DEVICE, "state reset failed");

  b->gdec.resize(C);
  HIPCHK(hipDeviceSynchronize());
  *out = b.release();
  return FMD_OK;
}

void fmd_batch_destroy(fmd_batch* b)
{
  delete b;
}

int fmd_batch_reset(fmd_batch* b)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  if (do_reset(b))
    return fail(FMD_ERR_DEVICE, "state reset failed");
  HIPCHK(hipDeviceSynchronize());
  return FMD_OK;
}

unsigned fmd_batch_channels(const fmd_batch* b)
{
  return b ? b->C : 0;
}

unsigned fmd_batch_max_audio_floats(const fmd_batch* b, unsigned samples)
{
  if (!b)
    return 0;
  const unsigned M = (samples + b->des.D - 1) / b->des.D + 1;
  return 2 * (unsigned(double(M) / double(b->des.rs_step)) + 4);
}

int fmd_batch_process_device(fmd_batch* b, const float* d_iq, size_t iq_channel_stride,
                             unsigned samples, float* d_audio, size_t audio_channel_stride,
                             unsigned* out_floats, void* stream_)
{
  if (!b || !d_iq || !d_audio)
    return fail(FMD_ERR_ARG, "fmd_batch_process_device: null argument");
  if (samples > FMD_MAX_BLOCK || samples < FMD_MIN_BLOCK)
    return fail(FMD_ERR_SIZE, "samples must be within [FMD_MIN_BLOCK, FMD_MAX_BLOCK]");
  const fmd::Design& d = b->des;
  const unsigned C = b->C, CP = b->CP, N = samples, D = d.D;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));

  /* ---- position plan (batch-uniform, mirrors the reference's bookkeeping) ---- */
  const unsigned pos = b->if_pos;
  const unsigned M = pos < N ? (N - pos + D - 1) / D : 0; // DownConvert.cpp:112,123
  if (M == 0)
    return fail(FMD_ERR_SIZE, "block shorter than the decimator phase");
  std::vector<unsigned> hb_in(d.hb.size());
  unsigned R = M;
  for (size_t s = 0; s < d.hb.size(); s++)
  {
    hb_in[s] = R;
    if (R < 2u * unsigned(d.hb[s].len - 1))
      return fail(FMD_ERR_SIZE, "block too short for the RDS half-band chain at this rate");
    R = (R + 1) / 2; // DownConvert.cpp:526
  }
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
  if (size_t(2) * A > audio_channel_stride && C > 1)
    return fail(FMD_ERR_ARG, "audio_channel_stride smaller than the audio produced");

  const unsigned T_lpf = unsigned(d.rds_lpf_taps.size());
  const unsigned T_mf = unsigned(d.rds_mf_taps.size());
  const unsigned T_alp = unsigned(d.lpf_taps.size());
  const unsigned Hbb = d.rs_order;
  b->call_index++;

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
    evset = &b->ev[size_t(b->prof_calls) * (ST_COUNT + 1)];
    b->prof_calls++;
  }
  auto mark = [&](int i) {
    if (evset && (b->profiling >= 2 || i <= 1))
      (void)hipEventRecord(evset[i], stream);
  };
  mark(0);

  /* ---- K1: tuner + IF decimating FIR ---- */
  {
    constexpr int TILE = 256;
    const unsigned ntiles = (M + TILE - 1) / TILE;
    const size_t lds = (size_t(TILE - 1) * D + d.if_order + 4) * sizeof(float2);
    if (lds > 160 * 1024)
      return fail(FMD_ERR_ARG, "IF filter window does not fit in LDS");
    const bool pow2 = (d.table_size & (d.table_size - 1)) == 0 && d.table_size <= 2 * TILE;
    // loads per lane needed to stage one tile in a single round trip (two samples per load)
    const unsigned rounds = ((size_t(TILE - 1) * D + d.if_order + 2) / 2 + TILE - 1) / TILE;
    auto kfn = &fmd::k_if_fir<TILE, 4, false>;
    if (pow2)
      kfn = rounds <= 2 ? &fmd::k_if_fir<TILE, 2, true>
          : rounds <= 4 ? &fmd::k_if_fir<TILE, 4, true>
          : rounds <= 6 ? &fmd::k_if_fir<TILE, 6, true>
                        : &fmd::k_if_fir<TILE, 8, true>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kfn, dim3(C * ntiles), dim3(TILE), lds, stream,
                       reinterpret_cast<const float2*>(d_iq), iq_channel_stride, N,
                       b->hist[b->hist_sel].p, b->hist[b->hist_sel ^ 1].p, b->lut.p, d.table_size,
                       b->lut_idx, b->if_coeff.p, d.if_order, D, pos, M, b->demod.p, b->Mstride,
                       ntiles, (C % 8 == 0) ? 1u : 0u);
  }
  mark(1);

  /* ---- K2: baseband-rate recurrences ---- */
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
    hipLaunchKernelGGL(fmd::k_if_level, dim3(C), dim3(64), 0, stream,
                       reinterpret_cast<const float2*>(d_iq), iq_channel_stride, N, b->lut.p,
                       d.table_size, b->lut_idx, b->st);
    hipLaunchKernelGGL(fmd::k_demod_serial, dim3(CP / 64), dim3(128), 0, stream, b->demod.p,
                       b->Mstride, M, C, CP, k, b->st, b->br.p, Hbb, b->mix.p,
                       unsigned(d.hb[0].len - 1), b->sctab.p,
                       FmdSincosTab{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo});
  }
  mark(2);

  /* ---- K3: half-band chain ---- */
  {
    const float2* in = b->mix.p;
    for (size_t s = 0; s < d.hb.size(); s++)
    {
      const unsigned n_out = (hb_in[s] + 1) / 2;
      const bool last = (s + 1 == d.hb.size());
      float2* outp = last ? b->rdsraw.p : b->hbbuf[s].p;
      const unsigned Hout = last ? (T_lpf - 1) : unsigned(d.hb[s + 1].len - 1);
      hipLaunchKernelGGL(fmd::k_halfband, dim3(CP / 64, (n_out + 4 * fmd::HB_R - 1) / (4 * fmd::HB_R)),
                         dim3(64, 4), 0, stream, in,
                         outp, n_out, d.hb[s].len, b->hbcoef[s], C, CP, Hout);
      in = outp;
    }
  }
  mark(3);

  /* ---- K4: RDS 75-tap low-pass ---- */
  hipLaunchKernelGGL(fmd::k_ring_fir<float2>, dim3(CP / 64, (R + fmd::RF_TI - 1) / fmd::RF_TI), dim3(64, 4),
                     size_t(T_lpf - 1 + fmd::RF_TI) * 64 * sizeof(float2), stream, b->rdsraw.p,
                     b->rlpf.p, R, int(T_lpf), b->rds_lpf_taps.p, b->rds_lpf_g, C, CP, 0u);
  mark(4);

  /* ---- K5: RDS recurrences and block sync ---- */
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
    k.mf_taps = int(T_mf);
    const FmdSincosTab sct{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo};
    hipLaunchKernelGGL(fmd::k_rds_pll, dim3(CP / 64), dim3(64), 0, stream, b->rlpf.p, R, C, CP, k,
                       b->st, b->rpll.p, T_mf - 1, b->sctab.p, sct);
    hipLaunchKernelGGL(fmd::k_ring_fir<float>, dim3(CP / 64, (R + fmd::RF_TI - 1) / fmd::RF_TI),
                       dim3(64, 4), size_t(T_mf - 1 + fmd::RF_TI) * 64 * sizeof(float), stream,
                       b->rpll.p, b->rmf.p, R, int(T_mf), b->mf_taps2.p, b->mf_g, C, CP, 0u);
    hipLaunchKernelGGL(fmd::k_rds_bits, dim3(CP / 64), dim3(64), 0, stream, b->rmf.p, R, C, CP, k,
                       b->st, b->call_index, b->queue.p, b->queue_count.p, b->queue_cap,
                       b->tap_sync.p, b->write_taps);
  }
  mark(5);

  /* ---- K6/K7: fractional resamplers (mono + stereo) ---- */
  hipLaunchKernelGGL(fmd::k_rs_table, dim3(A), dim3(64), 0, stream, b->rs_coeff.p, d.rs_order, p,
                     pstep, A, b->ktab.p, b->pidx.p);
  hipLaunchKernelGGL(fmd::k_resample, dim3(CP / 64, (A + 4 * fmd::RS_R - 1) / (4 * fmd::RS_R)),
                     dim3(64, 4), 0, stream, b->br.p, Hbb, d.rs_order, b->ktab.p, b->pidx.p, A,
                     b->rs.p, T_alp - 1, C, CP);
  mark(6);

  /* ---- audio 15 kHz low-pass on the (stereo, mono) pair ---- */
  hipLaunchKernelGGL(fmd::k_ring_fir<float2>, dim3(CP / 64, (A + fmd::RF_TI - 1) / fmd::RF_TI), dim3(64, 4),
                     size_t(T_alp - 1 + fmd::RF_TI) * 64 * sizeof(float2), stream, b->rs.p, b->alp.p, A,
                     int(T_alp), b->audio_taps.p, b->alpf_g, C, CP, 0u);
  mark(7);

  /* ---- K8: de-emphasis, notch, L/R ---- */
  {
    fmd::AudioConsts k{};
    k.de_alpha = d.de_alpha;
    k.n_b0 = d.notch.b0;
    k.n_b1 = d.notch.b1;
    k.n_b2 = d.notch.b2;
    k.n_a1 = d.notch.a1;
    k.n_a2 = d.notch.a2;
    hipLaunchKernelGGL(fmd::k_audio_tail, dim3(CP / 64), dim3(64), 0, stream, b->alp.p, A, C, CP, k,
                       b->st, d_audio, audio_channel_stride);
  }
  mark(8);

  /* ---- history rolls: keep the last H rows of every windowed buffer for the next call ---- */
  {
    const dim3 t(256);
    auto grid = [&](unsigned H) { return dim3((CP + 255) / 256, std::max(1u, std::min(H, 64u))); };
    hipLaunchKernelGGL(fmd::k_roll<float2>, grid(Hbb), t, 0, stream, b->br.p, Hbb, M, CP);
    const unsigned H0 = unsigned(d.hb[0].len - 1);
    hipLaunchKernelGGL(fmd::k_roll<float2>, grid(H0), t, 0, stream, b->mix.p, H0, hb_in[0], CP);
    for (size_t s = 1; s < d.hb.size(); s++)
    {
      const unsigned Hs = unsigned(d.hb[s].len - 1);
      hipLaunchKernelGGL(fmd::k_roll<float2>, grid(Hs), t, 0, stream, b->hbbuf[s - 1].p, Hs, hb_in[s], CP);
    }
    hipLaunchKernelGGL(fmd::k_roll<float2>, grid(T_lpf - 1), t, 0, stream, b->rdsraw.p, T_lpf - 1, R, CP);
    hipLaunchKernelGGL(fmd::k_roll<float>, grid(T_mf - 1), t, 0, stream, b->rpll.p, T_mf - 1, R, CP);
    hipLaunchKernelGGL(fmd::k_roll<float2>, grid(T_alp - 1), t, 0, stream, b->rs.p, T_alp - 1, A, CP);
  }
  mark(9);
  HIPCHK(hipGetLastError());

  /* ---- advance the host-tracked positions ---- */
  b->if_pos = pos + M * D - N;                      // DownConvert.cpp:132
  b->lut_idx = (b->lut_idx + N) % d.table_size;     // FmDecode.cpp:81
  b->rs_pos = new_rs_pos;                           // DownConvert.cpp:230-232
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

int fmd_batch_collect_rds(fmd_batch* b, fmd_rds_group* out, unsigned cap, int run_group_decoder,
                          void* stream_)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));
  unsigned n = 0;
  HIPCHK(hipMemcpyAsync(&n, b->queue_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (n > b->queue_cap)
    n = b->queue_cap; // overflow: the oldest queue_cap groups are kept
  std::vector<fmd::RdsGroupRec> recs(n);
  if (n)
    HIPCHK(hipMemcpyAsync(recs.data(), b->queue.p, size_t(n) * sizeof(fmd::RdsGroupRec),
                          hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemsetAsync(b->queue_count.p, 0, sizeof(unsigned), stream));
  HIPCHK(hipStreamSynchronize(stream));
  std::sort(recs.begin(), recs.end(), [](const fmd::RdsGroupRec& x, const fmd::RdsGroupRec& y) {
    if (x.call_index != y.call_index)
      return x.call_index < y.call_index;
    if (x.channel != y.channel)
      return x.channel < y.channel;
    return x.seq < y.seq;
  });
  unsigned k = 0;
  for (const auto& r : recs)
  {
    if (run_group_decoder && r.channel < b->C)
    {
      auto& g = b->gdec[r.channel];
      if (!g)
        g.reset(new fmd::GroupDecoder(&b->cb, b->user, r.channel));
      g->push(r.blocks);
    }
    if (out && k < cap)
    {
      out[k].channel = r.channel;
      out[k].call_index = r.call_index;
      for (int q = 0; q < 4; q++)
        out[k].blocks[q] = r.blocks[q];
      k++;
    }
  }
  return int(out ? k : n);
}

int fmd_batch_process_host(fmd_batch* b, const float* iq, size_t iq_channel_stride, unsigned samples,
                           float* audio, size_t audio_channel_stride, unsigned* out_floats)
{
  if (!b || !iq || !audio)
    return fail(FMD_ERR_ARG, "fmd_batch_process_host: null argument");
  HIPCHK(hipSetDevice(b->device));
  const unsigned C = b->C;
  const size_t dev_iq_stride = iq_channel_stride ? samples : 0;
  const size_t iq_floats = size_t(2) * samples * (iq_channel_stride ? C : 1);
  const size_t a_stride = (size_t(fmd_batch_max_audio_floats(b, samples)) + 3) & ~size_t(3);
  if (iq_floats > b->h_iq_cap)
  {
    b->h_iq.release();
    if (b->h_iq.alloc(iq_floats))
      return fail(FMD_ERR_DEVICE, "staging allocation failed");
    b->h_iq_cap = iq_floats;
  }
  if (a_stride * C > b->h_audio_cap)
  {
    b->h_audio.release();
    if (b->h_audio.alloc(a_stride * C))
      return fail(FMD_ERR_DEVICE, "staging allocation failed");
    b->h_audio_cap = a_stride * C;
  }
  if (iq_channel_stride)
    HIPCHK(hipMemcpy2D(b->h_iq.p, size_t(2) * samples * sizeof(float), iq,
                       size_t(2) * iq_channel_stride * sizeof(float), size_t(2) * samples * sizeof(float),
                       C, hipMemcpyHostToDevice));
  else
    HIPCHK(hipMemcpy(b->h_iq.p, iq, iq_floats * sizeof(float), hipMemcpyHostToDevice));
  unsigned nf = 0;
  int rc = fmd_batch_process_device(b, b->h_iq.p, dev_iq_stride, samples, b->h_audio.p, a_stride, &nf,
                                    nullptr);
  if (rc != FMD_OK)
    return rc;
  if (C > 1 && nf > audio_channel_stride)
    return fail(FMD_ERR_ARG, "audio_channel_stride smaller than the audio produced");
  HIPCHK(hipMemcpy2D(audio, (C > 1 ? audio_channel_stride : size_t(nf)) * sizeof(float), b->h_audio.p,
                     a_stride * sizeof(float), size_t(nf) * sizeof(float), C, hipMemcpyDeviceToHost));
  rc = fmd_batch_collect_rds(b, nullptr, 0, 1, nullptr);
  if (rc < 0)
    return rc;
  if (out_floats)
    *out_floats = nf;
  return FMD_OK;
}

int fmd_batch_get_status(fmd_batch* b, unsigned channel, fmd_status* stt)
{
  if (!b || !stt || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_get_status: bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  float if_level = 0, bb_mean = 0, bb_level = 0, p_level = 0;
  int stereo = 0, rstate = 0;
  HIPCHK(hipMemcpy(&if_level, b->st.F(fmd::F_IF_LEVEL) + channel, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&bb_mean, b->st.F(fmd::F_BB_MEAN) + channel, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&bb_level, b->st.F(fmd::F_BB_LEVEL) + channel, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&p_level, b->st.F(fmd::F_P_LEVEL) + channel, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&stereo, b->st.I(fmd::I_STEREO) + channel, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&rstate, b->st.I(fmd::I_R_STATE) + channel, 4, hipMemcpyDeviceToHost));
  stt->stereo_detected = stereo;
  // FmDecode.h:146-150
  const float tuned = float(-b->shifts[channel]) * b->des.fs_if / float(int(b->des.table_size));
  stt->tuning_offset = tuned + bb_mean * b->des.freq_dev;
  stt->interface_level = if_level;
  stt->baseband_level = bb_level;
  stt->pilot_level = 2 * p_level; // FmDecode.h:75
  stt->rds_state = rstate;
  return FMD_OK;
}

int fmd_batch_get_tap(fmd_batch* b, int tap, unsigned channel, float* out, unsigned cap_floats)
{
  if (!b || !out || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_get_tap: bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t CP = b->CP;
  const void* src = nullptr;
  size_t esize = 4, rows = 0, first_row = 0;
  const unsigned T_alp = unsigned(b->des.lpf_taps.size());
  switch (tap)
  {
    case FMD_TAP_DEMOD:
      if (size_t(2) * b->lastM > cap_floats)
        return fail(FMD_ERR_ARG, "tap buffer too small");
      HIPCHK(hipMemcpy(out, b->demod.p + size_t(channel) * b->Mstride, size_t(b->lastM) * 8,
                       hipMemcpyDeviceToHost));
      return int(b->lastM);
    /* windowed buffers were rolled at the end of the call: the block's rows are still in place
     * at [H, H+n) except the first H rows region, which now holds the tail -- read the data rows */
    case FMD_TAP_BASEBAND:
    case FMD_TAP_PILOT38:
    {
      rows = b->lastM;
      if (rows > cap_floats)
        return fail(FMD_ERR_ARG, "tap buffer too small");
      const char* s0 = reinterpret_cast<const char*>(b->br.p) +
                       (size_t(b->des.rs_order) * CP + channel) * 8 + (tap == FMD_TAP_PILOT38 ? 4 : 0);
      if (rows)
        HIPCHK(hipMemcpy2D(out, 4, s0, CP * 8, 4, rows, hipMemcpyDeviceToHost));
      return int(rows);
    }
    case FMD_TAP_MONO_RS:
    case FMD_TAP_STEREO_RS:
      src = reinterpret_cast<const float*>(b->rs.p) + (tap == FMD_TAP_MONO_RS ? 1 : 0);
      esize = 4;
      first_row = T_alp - 1;
      rows = b->lastA;
      {
        if (rows > cap_floats)
          return fail(FMD_ERR_ARG, "tap buffer too small");
        const char* s = reinterpret_cast<const char*>(src) + (first_row * CP + channel) * 8;
        HIPCHK(hipMemcpy2D(out, 4, s, CP * 8, 4, rows, hipMemcpyDeviceToHost));
        return int(rows);
      }
    case FMD_TAP_RDS_LPF:
      src = b->rlpf.p;
      esize = 8;
      rows = b->lastR;
      break;
    case FMD_TAP_RDS_PLL:
      src = b->rpll.p;
      first_row = b->des.rds_mf_taps.size() - 1;
      rows = b->lastR;
      break;
    case FMD_TAP_RDS_MF:
      src = b->rmf.p;
      rows = b->lastR;
      break;
    case FMD_TAP_RDS_SYNC:
      if (!b->write_taps)
        return fail(FMD_ERR_STATE, "RDS taps need fmd_batch_set_debug_taps(b, 1) before the call");
      src = b->tap_sync.p;
      rows = b->lastR;
      break;
    default:
      return fail(FMD_ERR_ARG, "unknown tap");
  }
  if (rows * (esize / 4) > cap_floats)
    return fail(FMD_ERR_ARG, "tap buffer too small");
  const char* s = reinterpret_cast<const char*>(src) + (first_row * CP + channel) * esize;
  if (rows)
    HIPCHK(hipMemcpy2D(out, esize, s, CP * esize, esize, rows, hipMemcpyDeviceToHost));
  return int(rows);
}

int fmd_batch_get_design(fmd_batch* b, int what, float* out, unsigned cap)
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "fmd_batch_get_design: bad argument");
  const fmd::Design& d = b->des;
  std::vector<float> v;
  switch (what)
  {
    case FMD_DESIGN_IF_TAPS:
      v = d.if_coeff;
      break;
    case FMD_DESIGN_RS_TAPS:
      v = d.rs_coeff;
      break;
    case FMD_DESIGN_AUDIO_LPF:
      v = d.lpf_taps;
      break;
    case FMD_DESIGN_RDS_LPF:
      v = d.rds_lpf_taps;
      break;
    case FMD_DESIGN_RDS_MF:
      v = d.rds_mf_taps;
      break;
    case FMD_DESIGN_LUT0:
      v = fmd::make_tuner_lut(d.table_size, b->shifts[0]);
      break;
    case FMD_DESIGN_SCALARS:
      v = {float(b->shifts[0]), d.demod_gain,  d.de_alpha,     d.pll_alpha,  d.pll_beta,
           d.nco_hl,            d.nco_ll,      d.p_minfreq,    d.p_maxfreq,  d.p_b0,
           d.p_a1,              d.p_a2,        d.p_lf_b0,      d.p_lf_b1,    d.p_freq0,
           float(d.p_lock_delay), float(d.rs_order), d.rs_step, d.rds_rate,  d.rds_nco_inc,
           d.rds_osc_cos,       d.rds_osc_sin, d.rds_pll_alpha, d.rds_pll_beta, d.rds_nco_hl,
           d.rds_nco_ll,        d.fs_bb,       float(d.rds_mf_taps.size()),
           d.notch.b0, d.notch.b1, d.notch.b2, d.notch.a1, d.notch.a2,
           d.bitsync.b0, d.bitsync.b1, d.bitsync.b2, d.bitsync.a1, d.bitsync.a2};
      break;
    default:
      return fail(FMD_ERR_ARG, "unknown design item");
  }
  for (size_t i = 0; i < v.size() && i < cap; i++)
    out[i] = v[i];
  return int(v.size());
}

int fmd_batch_set_debug_taps(fmd_batch* b, int enable)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  b->write_taps = enable != 0;
  return FMD_OK;
}

int fmd_batch_set_profiling(fmd_batch* b, int level)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  b->profiling = level < 0 ? 0 : (level > 2 ? 2 : level);
  b->prof_calls = 0; // restart the averaging window
  return FMD_OK;
}

int fmd_batch_get_stage_ms(fmd_batch* b, float* out, unsigned cap)
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "bad argument");
  if (!b->profiling || b->prof_calls == 0)
    return fail(FMD_ERR_STATE, "profiling not enabled or no call made since it was enabled");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  const int nst = b->profiling >= 2 ? ST_COUNT : 1;
  for (int i = 0; i < ST_COUNT; i++)
  {
    double sum = 0;
    if (i < nst)
      for (unsigned c = 0; c < b->prof_calls; c++)
      {
        float ms = 0;
        hipEvent_t* es = &b->ev[size_t(c) * (ST_COUNT + 1)];
        HIPCHK(hipEventElapsedTime(&ms, es[i], es[i + 1]));
        sum += ms;
      }
    if (unsigned(i) < cap)
      out[i] = i < nst ? float(sum / b->prof_calls) : -1.0f;
  }
  return int(b->prof_calls);
}

/* ---- single decoder = batch of one -------------------------------------------------------- */
struct fmd_decoder
{
  fmd_batch* b;
};

int fmd_create(const fmd_params* params, const fmd_callbacks* cb, void* user, fmd_decoder** out)
{
  if (!out)
    return fail(FMD_ERR_ARG, "null out");
  *out = nullptr;
  fmd_batch* b = nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  int rc = fmd_batch_create(params, 1, nullptr, dev, cb, user, &b);
  if (rc != FMD_OK)
    return rc;
  *out = new fmd_decoder{b};
  return FMD_OK;
}

void fmd_destroy(fmd_decoder* d)
{
  if (!d)
    return;
  fmd_batch_destroy(d->b);
  delete d;
}

int fmd_reset(fmd_decoder* d)
{
  return d ? fmd_batch_reset(d->b) : fail(FMD_ERR_ARG, "null decoder");
}

int fmd_process_stream(fmd_decoder* d, const float* iq, unsigned samples, float* audio)
{
  if (!d)
    return fail(FMD_ERR_ARG, "null decoder");
  unsigned nf = 0;
  int rc = fmd_batch_process_host(d->b, iq, 0, samples, audio, 0, &nf);
  return rc == FMD_OK ? int(nf) : rc;
}

int fmd_get_status(fmd_decoder* d, fmd_status* st)
{
  return d ? fmd_batch_get_status(d->b, 0, st) : fail(FMD_ERR_ARG, "null decoder");
}

/* ---- host-only pieces ----------------------------------------------------------------------- */
struct fmd_group_decoder
{
  fmd::GroupDecoder g;
  fmd_group_decoder(const fmd_callbacks* cb, void* user, unsigned ch) : g(cb, user, ch) {}
};

fmd_group_decoder* fmd_group_decoder_create(const fmd_callbacks* cb, void* user, unsigned channel)
{
  return new fmd_group_decoder(cb, user, channel);
}
void fmd_group_decoder_destroy(fmd_group_decoder* g)
{
  delete g;
}
void fmd_group_decoder_reset(fmd_group_decoder* g)
{
  if (g)
    g->g.reset();
}
void fmd_group_decoder_push(fmd_group_decoder* g, const uint16_t blocks[4])
{
  if (g)
    g->g.push(blocks);
}

int fmd_uecp_stuff_frame(const uint8_t* frame, unsigned len, uint8_t* out, unsigned cap)
{
  // cRadioReceiver::AddUECPDataFrame (RadioReceiver.cpp:387-414)
  unsigned k = 0;
  auto put = [&](uint8_t v) {
    if (k < cap)
      out[k] = v;
    k++;
  };
  put(0xFE);
  for (unsigned i = 0; i < len; i++)
  {
    const uint8_t v = frame[i];
    if (v < 0xFD)
      put(v);
    else
    {
      put(0xFD);
      put(uint8_t((v & 3) - 1));
    }
  }
  put(0xFF);
  return k <= cap ? int(k) : -1;
}

} // extern "C"
