/*
 * fmd_batch.hip -- C ABI (include/fmd.h) of the MI355X FM decoder: device memory, the
 * per-call position plan (host) and the kernel sequence.  gfx950 only, no CPU fallback.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared fmd_batch.hip -o libfmd_hip.so
 */
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fmd_design.hpp"
#include "fmd_groups.hpp"
#include "fmd_kernels.hip.h"
#include "fmd_receiver.hpp"

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
  unsigned min_samples = 0; // smallest call this geometry takes (fmd_batch_min_samples)
  int streams_sharing = 0;  // internal streams the create-time probe found sharing a hardware queue
  unsigned cpc = 1;         // channels per capture (fmd_batch_set_channels_per_capture): channel c reads capture c / cpc
  unsigned Mmax = 0, Mstride = 0, Amax = 0, Rmax = 0;
  std::vector<unsigned> hb_nmax; // max input length per HB stage
  // last call
  unsigned lastM = 0, lastA = 0, lastR = 0;

  // device memory
  DevBuf<float2> lut, hist[2], demod[2], br[2], mix[2], rdsraw[2], rlpf[2], rs[2], alp[2];
  std::vector<DevBuf<float2>> hbbuf; // input buffers of stages 1..n-1 (stage 0 reads mix)
  DevBuf<float> if_coeff, rs_coeff, rds_lpf_taps, mf_taps2, audio_taps, ktab;
  DevBuf<float> rpll, rmf, tap_sync;
  DevBuf<double> sctab, sctab256; // fmd_sincos_tab / fmd_sincos_p256 (the serial stage's two NCOs)
  DevBuf<int> pidx;
  static constexpr unsigned kBrFront = 32; // rows of zeros in front of the history rows (both resamplers reach below)
  float2* brp(int q) const { return br[q].p + size_t(kBrFront) * CP; } // first history row of br[q]
  unsigned rs_margin = 0, rs_row = 0; // ktab: zero entries around each output's taps, row length
  // k_resample_ring (large batches): outputs per wave (0 = this geometry does not fit the CU's LDS),
  // ring batches, batches per group in the tap table, row offset; plan buffers; -1 auto / 0 off / 1 on
  int rsr_R = 0, rsr_NW = 0;
  unsigned rsr_nbr = 0, rsr_nbm = 0;
  int rsr_rb = 0;
  DevBuf<float> rsr_tab;
  DevBuf<int> rsr_head, rsr_steps;
  int rsr_mode = -1;
  int n_cus = 256;
  // k_halfband_chain (large batches, the usual three-stage chains): step lists by (inputs, stretches),
  // uploaded the first time a call size is seen; the stages' last outputs on their way to the history
  // rows; -1 auto / 0 off / 1 on
  struct HbfPlan
  {
    unsigned n_in = 0, S = 0;
    DevBuf<fmd::HbStep> steps;
    DevBuf<int> seg_first;
  };
  std::vector<std::unique_ptr<HbfPlan>> hbf_plans;
  DevBuf<float2> hbf_tail1, hbf_tail2;
  int hbf_mode = -1;
  /* The RDS oscillator (CRDSDownConvert::ProcessData, DownConvert.cpp:436-442) as one sequence per batch,
   * for calls whose serial stage writes no mixed rows (k_demod_serial<.., MIX = false> +
   * k_halfband_chain<.., OSC>): a recurrence on its own state with an amplitude servo, independent of the
   * signal, started at (1, 0) in every decoder and advanced by every baseband sample.  The HOST computes a
   * call's M values while it submits the call (~60 us of one core; a lone GPU wave takes 0.37 ms for the
   * same dependent chain, on a stream the IF FIR needs), in page-locked staging (8 slots) that an
   * asynchronous copy takes to the device tables (4 in rotation: a call's table is read by its half-band
   * chain, two serial stages later), kOscH entries of history in front.  osc_on: every call (large batches
   * from creation; "halfband_chain" = 1 before the first call). */
  static constexpr unsigned kOscH = 64;
  DevBuf<float2> osc_tab[4];
  float2* h_osc = nullptr;       // [NSLOT][kOscH + Mmax + 8], page-locked
  hipEvent_t osc_ev[8] = {}; // [NSLOT] behind the copy out of a staging slot
  bool osc_ev_used[8] = {};
  size_t h_osc_stride = 0;
  float osc_re = 1.0f, osc_im = 0.0f; // CRDSDownConvert: m_Osc1 = (1, 0) (DownConvert.cpp:284)
  bool osc_on = false;
  int dbg_nomix = 1;
  // development switches (fmd_batch_debug_set; the library reads no environment variable)
  int dbg_fir_nt = 0;          // tiles per IF FIR workgroup (0: the library decides)
  int dbg_serial_claim = 0;    // whole-CU serial stage: every role wave claims its SIMD's register file
  int dbg_hb4 = 1, dbg_ring4 = 1; // 0: the generic half-band / ring-FIR kernels where the unrolled ones would run
  int dbg_fir_ro = 2;          // outputs per lane of the headline IF FIR form: 1 k_if_fir_mt, 2 / 3 k_if_fir_mt3
                               // (2: 288 600 MS/s and the FIR 0.975 ms inside the pipeline; 3: 287 500 and 1.00)
  int dbg_lpf_late = -1;       // stream layout of an overlapped call (process_device_impl): -1 the library decides,
                               // 0 low-pass filters on the heavy stream, 1 on a stream of their own, 2 at the
                               // heads of the light part's two streams
  // where a host-buffer call's time goes (fmd_batch_debug_host_ms): copy in, submission, wait + copy
  // out, RDS collection + group decoder callbacks; sums since the last query
  double host_ms[4] = {0, 0, 0, 0};
  unsigned host_calls = 0;
  DevBuf<long long> serial_probe; // FMD_SERIAL_PROBE=1: per-workgroup timing of the serial stage
  int dbg_stage_mask = 63;        // energy experiment: parts of a call that are launched (63 = all; else wrong results)
  DevBuf<float> fstate; // all float state arrays, CP each
  DevBuf<int> istate;
  DevBuf<uint16_t> r_data;
  static constexpr int NSLOT = 8; // event sets / RDS queues in rotation (call_index % NSLOT)
  static_assert(NSLOT == 8, "osc_ev / osc_ev_used above");
  DevBuf<fmd::RdsGroupRec> queue[NSLOT]; // never drained while a call that appends to it is in flight
  DevBuf<unsigned> queue_counts; // [NSLOT], contiguous: one copy reads them all
  unsigned* qcount(int q) const { return queue_counts.p + q; }
  // call whose groups were last taken out of the slot's queue (collect / export): a slot is not
  // visited again before a newer call has used it
  uint32_t drained_call[NSLOT] = {};
  // page-locked staging of fmd_batch_collect_rds: the counts and the records arrive by DMA, no
  // staging kernel, two synchronisations per collect
  unsigned* h_counts = nullptr;
  fmd::RdsGroupRec* h_recs = nullptr;
  size_t h_recs_cap = 0;
  unsigned queue_cap = 0;
  // fmd_batch_export_rds_device drains a queue asynchronously on the caller's stream: the event tells
  // the next call that appends to the same queue (NSLOT calls later) when it is empty
  hipEvent_t ev_drained[NSLOT] = {};
  bool drained_pending[NSLOT] = {};
  // running row count of one export (k_rds_export): a cursor per call out of a ring, so that two exports
  // on different streams never share one
  static constexpr unsigned kExportCursors = 16;
  DevBuf<unsigned> export_cursor;
  unsigned export_seq = 0;
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

  // Internal streams: the IF FIR of the next calls (s_fir), the serial stage of this call (s_ser), the heavy part of
  // the post chain behind it (s_post: half-band chain, resampler) and its light part (s_rds, s_lpf: see the stream
  // layouts in process_device_impl; a batch of one or two wavefronts runs its RDS chain and its audio chain side by
  // side on s_rds / s_post) are independent chains tied together, and to the caller's stream, with events.  Streams
  // of their own on purpose, and picked by measurement (pick_independent_streams): HIP multiplexes streams onto a
  // few hardware queues and two chains sharing a queue block each other.  demod, br, mix and the filters' inputs
  // are double-buffered by call parity so no chain waits on a buffer a younger call still reads.
  //   concurrency 0: everything on the caller's stream (also forced by profiling level 2)
  //   concurrency 1: internal streams, the caller's stream is ordered after every call (default)
  //   concurrency 2: as 1, but the caller's stream is only ordered after a call by fmd_batch_wait /
  //                  fmd_batch_collect_rds; lets call k+1's FIR overlap call k's serial stages
  int concurrency = 1;
  hipStream_t s_fir = nullptr, s_ser = nullptr, s_post = nullptr, s_rds = nullptr, s_lpf = nullptr;
  // What the light part of a call's post chain (RDS PLL, matched filter, bit recovery, audio tail) needs to know
  // about its call: see launch_light_rds / launch_light_audio.
  struct LightJob
  {
    unsigned R = 0, A = 0, mf_g = 0, alpf_g = 0;
    unsigned rds_lpf_g = 0;            // ring phase of the RDS low-pass at this call
    int q = 0, es = 0, sq = 0;
    bool events = true;                // false: everything on the caller's stream, no event is recorded or waited for
    bool lpf_here = false;             // the two complex low-pass filters are part of the light part (layout 2)
    int audio_after = -1;              // event of the call the audio half waits for on its stream (-1: stream order)
    bool status_after_rds = false;     // the two halves are on different streams: the status record waits for EV_RDS
    hipEvent_t prev_aud = nullptr;     // ... and the bit recovery for the previous call's status record
    uint32_t call_index = 0;
    float* d_audio = nullptr;
    size_t audio_stride = 0;
    hipEvent_t tl0 = nullptr, tl1 = nullptr; // profiling level 1: the audio tail's own start / stop
  };
  /* More channels than the whole-CU pipeline is built for (kSubBatchChannels = 64 CUs' worth of serial stage): the
   * batch the caller holds is a SHELL over ceil(C / 8192) sub-batches of equal size, each a complete batch of its own
   * (buffers, channel state, RDS queues, status snapshot), all on the shell's five streams.  A call is submitted
   * sub-batch by sub-batch, so that on the streams the sub-batch calls form ONE sequence of 8192-channel calls --
   * FIR(k, s + 1) behind heavy(k, s - 1), serial stages back to back -- i.e. the pipeline DESIGN.md section 2
   * describes, at its rate, whatever C is (channels are independent: FmDecode.h:201-212).  The shell owns the
   * streams, the caller-facing bookkeeping (group decoders, host staging, export cursor) and nothing on the device. */
  std::vector<std::unique_ptr<fmd_batch>> subs;
  std::vector<unsigned> sub_ch0;   // first channel of every sub-batch, and C behind the last
  bool owns_streams = true;        // false: a sub-batch, the streams are its shell's
  hipEvent_t vheavy[2] = {nullptr, nullptr}; // shell: EV_HEAVY of the last two sub-batch calls of the common sequence
  // sub-batch: EV_HEAVY of the sub-batch call two back in the common sequence -- this call's IF FIR stays behind it
  // (process_device_impl); else null
  hipEvent_t sched_prev2 = nullptr;
  bool split_post = false;
  bool serial_exclusive = false; // serial stage owns whole CUs (small batches, see the launch)
  enum { EV_IN, EV_FIR, EV_INDONE, EV_SER, EV_AUD, EV_RDS, EV_HEAVY, EV_RDSH, EV_ALP, EV_DEC, EV_ROLL, EV_N };
  hipEvent_t cev[NSLOT][EV_N] = {};
  bool cev_ready = false;
  uint32_t slot_call[NSLOT] = {}; // call index that last used the slot (0 = never)
  std::vector<hipEvent_t> ev; // [calls][ST_COUNT + 1]
  unsigned prof_calls = 0;

  // Device-side error word (fmd::DevErr bits) in host-mapped memory: kernels OR into it, the host
  // reads it without a copy or a synchronisation.  `failed`: a call broke off after its first launch
  // or a kernel reported an error -- the channel state is no longer trustworthy, every later call is
  // refused until fmd_batch_reset / destroy.
  unsigned* h_err = nullptr; // [0] fatal bits, [1] recoverable ones (groups lost), see fmd::DevErr
  bool failed = false;
  std::string fail_msg;
  unsigned spin_limit = 1u << 20; // fmd_batch_debug_set_spin_limit
  bool if_dry_run = false;        // launch_if_stage stops behind its feasibility checks
  // Status snapshot in host-mapped memory, written by the last kernel of every call
  // (fmd::HostStatusWord): what the getters read -- no device call, no batch bookkeeping touched,
  // so they are safe from any thread while another one is inside a process call.
  unsigned* h_status = nullptr;
  DevBuf<unsigned> d_status; // the same record in device memory: what the kernels write (k_status_publish copies)
  unsigned host_seq = 0; // tags of snapshot updates made by the host (create, reset)

  ~fmd_batch()
  {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    subs.clear(); // (sub-batches first: the streams they run on are this object's)
    lut.release();
    hist[0].release();
    hist[1].release();
    demod[0].release();
    demod[1].release();
    mix[0].release();
    mix[1].release();
    rdsraw[0].release();
    rdsraw[1].release();
    rlpf[0].release();
    rlpf[1].release();
    rs[0].release();
    rs[1].release();
    alp[0].release();
    alp[1].release();
    for (auto& b : hbbuf)
      b.release();
    if_coeff.release();
    rs_coeff.release();
    br[0].release();
    br[1].release();
    rds_lpf_taps.release();
    mf_taps2.release();
    audio_taps.release();
    ktab.release();
    rsr_tab.release();
    for (auto& pl : hbf_plans)
    {
      pl->steps.release();
      pl->seg_first.release();
    }
    hbf_tail1.release();
    hbf_tail2.release();
    for (auto& t : osc_tab)
      t.release();
    if (h_osc)
      (void)hipHostFree(h_osc);
    for (auto e : osc_ev)
      if (e)
        (void)hipEventDestroy(e);
    rsr_head.release();
    rsr_steps.release();
    rpll.release();
    rmf.release();
    tap_sync.release();
    sctab.release();
    sctab256.release();
    pidx.release();
    serial_probe.release();
    fstate.release();
    istate.release();
    r_data.release();
    for (int q = 0; q < NSLOT; q++)
    {
      queue[q].release();
      if (ev_drained[q])
        (void)hipEventDestroy(ev_drained[q]);
    }
    export_cursor.release();
    queue_counts.release();
    if (h_counts)
      (void)hipHostFree(h_counts);
    if (h_recs)
      (void)hipHostFree(h_recs);
    if (cev_ready)
      for (auto& row : cev)
        for (auto& e : row)
          (void)hipEventDestroy(e);
    if (owns_streams)
      for (hipStream_t st : {s_fir, s_ser, s_post, s_rds, s_lpf})
        if (st)
          (void)hipStreamDestroy(st);
    h_iq.release();
    h_audio.release();
    if (h_err)
      (void)hipHostFree(h_err);
    if (h_status)
      (void)hipHostFree(h_status);
    d_status.release();
    for (auto& e : ev)
      (void)hipEventDestroy(e);
  }
};

namespace
{

constexpr unsigned kMaxProfCalls = 512;
// channels of one batch with buffers of its own: 64 CUs' worth of the whole-CU serial stage (128 channels per CU), the
// size the pipeline is balanced for.  (Measured, round 6: ONE batch of 8320 channels = 65 CUs still runs at the
// 8192-channel rate, 8448 / 8704 / 9216 lose 10-13 % -- profiles/r6_f_*: no headroom above it worth a special case.)
constexpr unsigned kSubBatchChannels = 8192;

inline bool is_shell(const fmd_batch* b)
{
  return !b->subs.empty();
}

/* the sub-batch that owns a channel, and the channel's index there (a plain batch: itself) */
inline fmd_batch* owner_of(fmd_batch* b, unsigned channel, unsigned* local)
{
  if (!is_shell(b))
  {
    *local = channel;
    return b;
  }
  size_t s = 0;
  while (s + 1 < b->subs.size() && channel >= b->sub_ch0[s + 1])
    s++;
  *local = channel - b->sub_ch0[s];
  return b->subs[s].get();
}

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
  void* derr = nullptr;
  if (b->h_err && hipHostGetDevicePointer(&derr, b->h_err, 0) == hipSuccess)
    b->st.err = static_cast<unsigned*>(derr);
  void* dhs = nullptr;
  if (b->h_status && hipHostGetDevicePointer(&dhs, b->h_status, 0) == hipSuccess)
    b->st.hs = static_cast<unsigned*>(dhs);
  b->st.ds = b->d_status.p;
  // bound of the serial stage's LDS hand-off waits (~0.1 s; fmd_batch_debug_set_spin_limit)
  b->st.spin_limit = b->spin_limit;
}

/* The host's own updates of the status snapshot (initial values, Reset): same protocol as the
 * kernel's (fmd::HostStatusWord), with tags no call index ever has.  Only called while no call is in
 * flight. */
void host_status_update(fmd_batch* b, const std::function<void(unsigned* rec, size_t stride)>& edit)
{
  const size_t CP = b->CP;
  const unsigned tag = 0x80000000u | ++b->host_seq;
  for (unsigned c = 0; c < b->C; c++)
  {
    unsigned* h = b->h_status + c;
    __atomic_store_n(&h[fmd::HS_SEQ_BEGIN * CP], tag, __ATOMIC_RELEASE);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    edit(h, CP);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    __atomic_store_n(&h[fmd::HS_SEQ_END * CP], tag, __ATOMIC_RELEASE);
  }
}

/* One channel's snapshot, consistent (taken between two equal sequence words). */
bool host_status_read(const fmd_batch* b, unsigned channel, unsigned out[fmd::HS_WORDS])
{
  const size_t CP = b->CP;
  const unsigned* h = b->h_status + channel;
  for (int tries = 0; tries < 100000; tries++)
  {
    const unsigned e = __atomic_load_n(&h[fmd::HS_SEQ_END * CP], __ATOMIC_ACQUIRE);
    for (int w = fmd::HS_SEQ_BEGIN + 1; w < fmd::HS_SEQ_END; w++)
      out[w] = __atomic_load_n(&h[size_t(w) * CP], __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const unsigned g = __atomic_load_n(&h[fmd::HS_SEQ_BEGIN * CP], __ATOMIC_ACQUIRE);
    if (e == g)
    {
      out[fmd::HS_SEQ_BEGIN] = out[fmd::HS_SEQ_END] = e;
      return true;
    }
  }
  return false; // a writer that never finishes: only a hung device
}

/* A step that cannot report through its return value (stream bookkeeping inside a launch helper)
 * failed: the batch refuses further calls (see check_device_errors). */
void mark_failed(fmd_batch* b, const char* what)
{
  if (!b->failed)
  {
    b->failed = true;
    b->fail_msg = what;
  }
}

/* Turns the device-side error word, and an earlier broken-off call, into an error code. */
int check_device_errors(fmd_batch* b)
{
  if (!b->subs.empty())
  { // a shell: its sub-batches' words; one that failed fails the whole batch
    for (auto& sb : b->subs)
      if (check_device_errors(sb.get()) != FMD_OK && !b->failed)
      {
        b->failed = true;
        b->fail_msg = sb->fail_msg;
      }
    return b->failed ? fail(FMD_ERR_DEVICE, b->fail_msg) : FMD_OK;
  }
  const unsigned e = b->h_err ? __atomic_load_n(&b->h_err[0], __ATOMIC_ACQUIRE) : 0u;
  if (e && !b->failed)
  {
    b->failed = true;
    b->fail_msg = "device-side error:";
    if (e & fmd::DEVERR_SERIAL_HANDSHAKE)
      b->fail_msg += " serial stage hand-off timed out (results of that call are invalid)";
  }
  if (b->failed)
    return fail(FMD_ERR_DEVICE, b->fail_msg);
  return FMD_OK;
}

/* RDS groups that did not fit a queue or the caller's record buffer are lost, nothing else: audio and
 * channel state are intact and the batch stays usable.  Reported once (FMD_WARN_RDS_LOST), then
 * cleared (atomically: a kernel setting the flag again at that moment is seen by the next query). */
int take_lost_groups(fmd_batch* b)
{
  if (!b->subs.empty())
  {
    int rc = FMD_OK;
    for (auto& sb : b->subs)
      if (take_lost_groups(sb.get()) != FMD_OK)
        rc = FMD_WARN_RDS_LOST;
    return rc;
  }
  // one exchange: a kernel that ORs the flag in between a load and a store would have its loss wiped
  if (!b->h_err || !__atomic_exchange_n(&b->h_err[1], 0u, __ATOMIC_ACQ_REL))
    return FMD_OK;
  g_err = "RDS groups were lost: a call's group queue or the export buffer was full (drain every call's "
          "groups with fmd_batch_collect_rds / fmd_batch_export_rds_device, with cap >= the groups queued)";
  return FMD_WARN_RDS_LOST;
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
template <class IN>
int launch_if_stage(fmd_batch* b, const void* d_iq, size_t iq_channel_stride, unsigned N, unsigned pos,
                    unsigned M, int q, hipStream_t sF, const std::function<void(int)>& mark,
                    hipEvent_t ev_start, hipEvent_t ev_stop);

int do_reset(fmd_batch* b)
{
  if (!b->subs.empty())
  {
    for (auto& sb : b->subs)
      if (do_reset(sb.get()))
        return -1;
    for (auto& g : b->gdec)
      if (g)
        g->reset();
    b->failed = false;
    b->fail_msg.clear();
    return 0;
  }
  const size_t CP = b->CP;
  using namespace fmd;
  const ChannelState& s = b->st;
  const int fz[] = {F_IF_LEVEL, F_BB_MEAN, F_BB_LEVEL, F_DC_OFF, F_NCO_INCR, F_NCO_PHASE, F_R_PHASE,
                    F_R_FREQ, F_R_W1, F_R_W2, F_R_LAST_SYNC, F_R_LAST_SLOPE, F_R_LAST_DATA};
  for (int slot : fz)
    if (hipMemset(s.F(slot), 0, CP * sizeof(float)) != hipSuccess)
      return -1;
  const int iz[] = {I_STEREO, I_STEREO_Q0, I_STEREO_Q1, I_STEREO_Q2, I_STEREO_Q3, I_R_LAST_BIT, I_R_BITPOS, I_R_BLOCK, I_R_STATE, I_R_BOFF};
  for (int slot : iz)
    if (hipMemset(s.I(slot), 0, CP * sizeof(int)) != hipSuccess)
      return -1;
  // RDS LPF ring (history rows of rdsraw), matched filter ring, positions
  if (zero_rows(b->rdsraw[0].p, b->des.rds_lpf_taps.size() - 1, CP) ||
      zero_rows(b->rdsraw[1].p, b->des.rds_lpf_taps.size() - 1, CP))
    return -1;
  if (zero_rows(b->rpll.p, b->des.rds_mf_taps.size() - 1, CP))
    return -1;
  b->rds_lpf_g = 0;
  b->mf_g = 0;
  for (auto& g : b->gdec)
    if (g)
      g->reset();
  if (b->h_err)
  {
    __atomic_store_n(&b->h_err[0], 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&b->h_err[1], 0u, __ATOMIC_RELEASE);
  }
  b->failed = false;
  b->fail_msg.clear();
  // the getters' snapshot follows: the meters Reset clears (FmDecode.cpp:326-338) read zero, pilot
  // level and the receiver's audio meter stay
  if (b->d_status.p)
    for (int w : {fmd::HS_IF_LEVEL, fmd::HS_BB_MEAN, fmd::HS_BB_LEVEL, fmd::HS_STEREO, fmd::HS_R_STATE})
      if (hipMemset(b->d_status.p + size_t(w) * CP, 0, CP * sizeof(unsigned)) != hipSuccess)
        return -1;
  if (b->h_status)
    host_status_update(b, [](unsigned* h, size_t CP) {
      for (int w : {fmd::HS_IF_LEVEL, fmd::HS_BB_MEAN, fmd::HS_BB_LEVEL, fmd::HS_STEREO, fmd::HS_R_STATE})
        __atomic_store_n(&h[size_t(w) * CP], 0u, __ATOMIC_RELAXED);
    });
  return 0;
}

/* HIP multiplexes streams onto a few hardware queues, and two of this library's chains on one
 * queue block each other (the FIR of the next call would queue behind the previous call's post
 * chain: measured -13 % when just one unrelated stream created earlier shifted the assignment).
 * So the internal streams are picked by measurement: candidates are created until `n` of them run
 * a no-op kernel at once while all already chosen ones are kept busy by a spinning wave. */
int pick_independent_streams(int n, const int* priority, hipStream_t* out, int* n_sharing)
{ // *n_sharing: how many of the n streams had to be taken although the probe found them behind another one
  std::vector<hipStream_t> rejected;
  int have = 0;
  for (int attempt = 0; attempt < 24 && have < n; attempt++)
  {
    hipStream_t cand = nullptr;
    if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, priority[have]) != hipSuccess)
      break;
    bool independent = true;
    {
      const long long spin = 1200000; // ~0.5 ms
      for (int i = 0; i < have; i++)
        hipLaunchKernelGGL(fmd::k_probe_spin, dim3(1), dim3(64), 0, out[i], spin, nullptr);
      // the default stream too: it is the caller's stream more often than not
      hipLaunchKernelGGL(fmd::k_probe_spin, dim3(1), dim3(64), 0, nullptr, spin, nullptr);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(fmd::k_probe_nop, dim3(1), dim3(64), 0, cand, nullptr);
      (void)hipStreamSynchronize(cand);
      const double us =
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      independent = us < 250.0; // behind a spinning wave it would take ~500 us
      for (int i = 0; i < have; i++)
        (void)hipStreamSynchronize(out[i]);
      (void)hipStreamSynchronize(nullptr);
    }
    if (independent)
      out[have++] = cand;
    else
      rejected.push_back(cand);
  }
  // whatever is still missing (no independent queue left): take the rejected ones, it still works
  *n_sharing = 0;
  while (have < n && !rejected.empty())
  {
    out[have++] = rejected.back();
    rejected.pop_back();
    ++*n_sharing;
  }
  for (hipStream_t s : rejected)
    (void)hipStreamDestroy(s);
  return have == n ? 0 : -1;
}

/* The steps of k_halfband_chain for a call of n_in inputs cut into S stretches of the last stage's
 * outputs (see the kernel): per step how far each stage runs -- at most 16 / 8 / 4 outputs, no further
 * than its input ring has rows for, and never onto a ring row the next stage still needs. */
fmd_batch::HbfPlan* hbf_plan(fmd_batch* b, unsigned n_in, unsigned S)
{
  for (auto& pl : b->hbf_plans)
    if (pl->n_in == n_in && pl->S == S)
      return pl.get();
  const fmd::Design& d = b->des;
  const int L1H = d.hb[1].len - 1, L2H = d.hb[2].len - 1, RING = fmd::HBF_RING;
  const int n0 = int(n_in + 1) / 2, n1 = (n0 + 1) / 2, n2 = (n1 + 1) / 2;
  std::vector<fmd::HbStep> steps;
  std::vector<int> first;
  const int per = (n2 + int(S) - 1) / int(S);
  for (int a = 0; a < n2; a += per)
  {
    first.push_back(int(steps.size()));
    const int e = std::min(n2, a + per);
    // what the stretch's outputs need of stages 1 and 0; the call's last stretch also computes the outputs
    // behind that (they are part of the delay lines the next call starts from)
    const bool last = e == n2;
    const int need1 = last ? n1 : 2 * (e - 1) + 1, need0 = last ? n0 : 2 * (need1 - 1) + 1;
    int d2 = a, d1 = std::max(0, 2 * a - L2H), d0 = std::max(0, 2 * d1 - L1H);
    while (d2 < e || d1 < need1 || d0 < need0)
    {
      const int a_hi = std::max(d0, std::min({d0 + 16, need0, 2 * d1 - L1H + RING}));
      const int b_hi = std::max(d1, std::min({d1 + 8, need1, a_hi > 0 ? (a_hi - 1) / 2 + 1 : 0, 2 * d2 - L2H + RING}));
      const int c_hi = std::max(d2, std::min({d2 + 4, e, b_hi > 0 ? (b_hi - 1) / 2 + 1 : 0}));
      if (a_hi == d0 && b_hi == d1 && c_hi == d2)
        return nullptr; // cannot happen: a stage can always move
      steps.push_back(fmd::HbStep{d0, a_hi - d0, d1, b_hi - d1, d2, c_hi - d2, 0, 0});
      d0 = a_hi;
      d1 = b_hi;
      d2 = c_hi;
    }
    while ((steps.size() - size_t(first.back())) % 4) // the kernel takes a stretch's steps four at a time
      steps.push_back(fmd::HbStep{d0, 0, d1, 0, d2, 0, 0, 0});
  }
  first.push_back(int(steps.size()));
  std::unique_ptr<fmd_batch::HbfPlan> pl(new fmd_batch::HbfPlan);
  pl->n_in = n_in;
  pl->S = unsigned(first.size() - 1);
  if (pl->steps.alloc(steps.size()) || pl->seg_first.alloc(first.size()) ||
      upload(pl->steps.p, steps.data(), steps.size() * sizeof(fmd::HbStep)) ||
      upload(pl->seg_first.p, first.data(), first.size() * sizeof(int)))
    return nullptr;
  if (b->hbf_plans.size() >= 16) // call sizes keep changing: forget the oldest list
  {
    (void)hipDeviceSynchronize();
    b->hbf_plans.front()->steps.release();
    b->hbf_plans.front()->seg_first.release();
    b->hbf_plans.erase(b->hbf_plans.begin());
  }
  b->hbf_plans.push_back(std::move(pl));
  return b->hbf_plans.back().get();
}

/* k_resample_ring's geometry for one of its forms (outputs per wave x waves): does a step's window
 * (+ alignment, + the even-count batch) fit 39 ring batches of 4 KB, and a step's new rows the
 * registers that carry them?  Leaves rsr_R = 0 when not. */
bool rsr_configure(fmd_batch* b, int form)
{
  // 2 x 8 first: two waves per SIMD cover each other's waits (0.30 ms alone at 8192 channels against 0.36
  // for 4 x 4; 2 x 4, half the outputs per step, is what longer windows still fit: 0.87 ms)
  static const int forms[3][2] = {{2, 8}, {4, 4}, {2, 4}};
  const fmd::Design& d = b->des;
  const int R = forms[form][0], NW = forms[form][1];
  const double step = double(d.rs_step);
  const unsigned per_step = unsigned(NW * R);
  const unsigned span = d.rs_order + 1u + unsigned(std::ceil((per_step - 1) * step)) + 1u;
  const unsigned nbr = (span - 1u + 7u) / 8u + 2u;
  const unsigned new_rows = unsigned(std::ceil(per_step * step)) + 8u;
  b->rsr_R = 0;
  if (nbr > 39u || new_rows > unsigned(fmd::RSR_ROWS))
    return false;
  b->rsr_R = R;
  b->rsr_NW = NW;
  b->rsr_nbr = nbr;
  b->rsr_rb = int((d.rs_order + 7u) / 8u * 8u + 8u);
  const unsigned gspan = d.rs_order + 1u + unsigned(std::ceil((R - 1) * step)) + 1u;
  b->rsr_nbm = (gspan - 1u + 7u) / 8u + 2u + 1u; // + 1: the walk's dummy last load
  return true;
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

} // extern "C"

namespace
{

/* The internal streams of a batch: the FIR feeds the pipeline and is the bandwidth-bound kernel -- dispatch it
 * first; the post chain has slack every call and goes last. */
int create_streams(fmd_batch* b)
{
  int lo = 0, hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi)); // lo = least, hi = greatest priority
  // (the two heavy chains side by side on a fifth stream: measured slower in rounds 1-3, removed.)  The
  // fifth stream carries the post chain's two low-pass filters or the light part's audio half: see the stream
  // layouts in process_device_impl
  const int nstreams = 5;
  // (the light chain's stream at the high priority too: measured twice, no difference; the heavy part's,
  // which since round 4 is on the loop that closes the period: 279 100 against 279 200 MS/s; with the
  // low-pass filters' as well: -1.6 %)
  const int prio[5] = {hi, hi, lo, lo, lo};
  hipStream_t st4[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (pick_independent_streams(nstreams, prio, st4, &b->streams_sharing) != 0)
    return fail(FMD_ERR_DEVICE, "could not create the internal streams");
  b->s_fir = st4[0];
  b->s_ser = st4[1];
  b->s_post = st4[2];
  b->s_rds = st4[3];
  b->s_lpf = st4[4];
  return FMD_OK;
}

/* One batch with buffers and state of its own: what fmd_batch_create returns up to kSubBatchChannels channels, and
 * every sub-batch of a larger one (shell != null: on the shell's streams). */
int create_one(const fmd_params* params, unsigned n_channels, const int* tuning_shifts, int device,
               const fmd_callbacks* cb, void* user, fmd_batch* shell, fmd_batch** out)
{
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
  if (params->fir_reduction != FMD_FIR_SEQUENTIAL && params->fir_reduction != FMD_FIR_SHUFFLE_PARITY_WAIVED &&
      params->fir_reduction != FMD_FIR_FMA_PARITY_WAIVED)
    return fail(FMD_ERR_ARG, "fmd_batch_create: fir_reduction must be 0 (sequential, the parity mode), "
                             "FMD_FIR_FMA_PARITY_WAIVED (fused multiply-add in the reference's tap order) or "
                             "FMD_FIR_SHUFFLE_PARITY_WAIVED (shuffle-reduced: 1.2e-5 RMS from the reference, "
                             "outside the 1e-5 contract)");
  b->params.fir_reduction = params->fir_reduction == FMD_FIR_SHUFFLE_PARITY_WAIVED ? 1
                            : params->fir_reduction == FMD_FIR_FMA_PARITY_WAIVED   ? 2
                                                                                   : 0;
  try
  {
    b->des = fmd::make_design(b->params);
  }
  catch (const std::exception& e)
  {
    return fail(FMD_ERR_ARG, e.what());
  }
  const fmd::Design& d = b->des;
  { // Smallest call.  Short blocks are decoded like the reference decodes them (a half-band stage
    // below L inputs passes its input on unfiltered, below 2 (L - 1) it refills its delay line from
    // outputs, a block shorter than a filter keeps part of the old history: process_device_impl);
    // what is refused are only blocks so short that a stage would be left with NO sample -- there the
    // reference divides by zero in its level meters (FmDecode.cpp:522-539, RadioReceiver.cpp:584-598)
    // -- and blocks that give the unrolled 11-tap class fewer inputs than it reads unconditionally.
    // The bound holds in every phase of the decimator: baseband samples M >= N / D rounded down.
    unsigned mmin = 5; // an audio frame falls into every block of >= ceil(step) + 1 baseband samples
    while (double(mmin) < double(d.rs_step) + 1.0)
      mmin++;
    for (;; mmin++)
    {
      unsigned n = mmin;
      bool ok = true;
      for (const auto& h : d.hb)
      {
        if (h.cic)
        { // (an even number of inputs: process_device_impl)
          ok = ok && n >= 2 && n % 2 == 0;
          n = n / 2;
        }
        else if (h.len == 11)
        {
          ok = ok && n >= 20;
          n = n / 2;
        }
        else
          n = n < unsigned(h.len) ? n / 2 : (n + 1) / 2;
      }
      if (ok && n >= 1)
        break;
    }
    const unsigned long long need = 1ull * mmin * d.D;
    if (need > FMD_MAX_BLOCK)
      return fail(FMD_ERR_ARG, "this geometry needs calls longer than the largest block (65536 samples)");
    b->min_samples = unsigned(need);
  }
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

  /* The serial stage addresses its input rows (demod) and its two output row buffers with 32-bit
   * byte offsets per lane (fmd_kernels.hip.h, k_demod_serial): a batch stays below 4 GB in each.  At
   * 2.4 MS/s that is ~80 000 channels; at 400 kS/s (no decimation) 8 000. */
  if (size_t(b->Mstride) * C * sizeof(float2) + 4096 >= (size_t(1) << 32) ||
      size_t(b->Mmax) * CP * sizeof(float2) >= (size_t(1) << 32))
    return fail(FMD_ERR_ARG, "too many channels for one batch at this sample rate (4 GB per row buffer): split the batch");
  int bad = 0;
  bad |= b->lut.alloc(size_t(d.table_size) * C);
  bad |= b->hist[0].alloc(size_t(d.if_order) * C);
  bad |= b->hist[1].alloc(size_t(d.if_order) * C);
  // + DS: the serial stage loads whole chunks, also behind the last sample of the last channel
  bad |= b->demod[0].alloc(size_t(b->Mstride) * C + fmd::DS);
  bad |= b->demod[1].alloc(size_t(b->Mstride) * C + fmd::DS);
  bad |= b->if_coeff.alloc(d.if_coeff.size() + 64); // zero padding: fir_long_e1_asm's dummy load
  bad |= b->rs_coeff.alloc(d.rs_coeff.size());
  // rows of zeros in front: the resamplers' last batch reaches below the window (k_resample: RS_B rows,
  // k_resample_ring: down to the batch border below, + one batch); 8 rows behind: its top batch
  static_assert(fmd_batch::kBrFront >= 2 * fmd::RS_B, "front rows of br");
  bad |= b->br[0].alloc(size_t(fmd_batch::kBrFront + d.rs_order + b->Mmax + 8) * CP);
  bad |= b->br[1].alloc(size_t(fmd_batch::kBrFront + d.rs_order + b->Mmax + 8) * CP);
  if (d.hb.empty())
    return fail(FMD_ERR_ARG, "baseband rate too low for the RDS decimation chain");
  bad |= b->mix[0].alloc(size_t(d.hb[0].len - 1 + b->Mmax) * CP);
  bad |= b->mix[1].alloc(size_t(d.hb[0].len - 1 + b->Mmax) * CP);
  b->hbbuf.resize(d.hb.size() - 1);
  for (size_t s = 1; s < d.hb.size(); s++)
    bad |= b->hbbuf[s - 1].alloc(size_t(d.hb[s].len - 1 + b->hb_nmax[s]) * CP);
  if (d.hb.size() == 3)
  {
    bad |= b->hbf_tail1.alloc(size_t(d.hb[1].len - 1) * CP);
    bad |= b->hbf_tail2.alloc(size_t(d.hb[2].len - 1) * CP);
    for (auto& t : b->osc_tab)
      bad |= t.alloc(size_t(fmd_batch::kOscH) + b->Mmax + 8);
    b->h_osc_stride = size_t(fmd_batch::kOscH) + b->Mmax + 8;
    bad |= hipHostMalloc(reinterpret_cast<void**>(&b->h_osc), fmd_batch::NSLOT * b->h_osc_stride * sizeof(float2),
                         hipHostMallocDefault) != hipSuccess;
    if (!bad)
      std::memset(b->h_osc, 0, fmd_batch::NSLOT * b->h_osc_stride * sizeof(float2));
    for (auto& e : b->osc_ev)
      bad |= hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess;
    // the batch-wide oscillator sequence is only read by k_halfband_chain, which exists for two chains
    const int h0 = (d.hb[0].len - 1) / 2, h1 = (d.hb[1].len - 1) / 2, h2 = (d.hb[2].len - 1) / 2;
    const bool chain_kind = h0 == 7 && ((h1 == 11 && h2 == 21) || (h1 == 9 && h2 == 17));
    b->osc_on = chain_kind && CP / 64 >= 64 && d.hb[0].len - 1 <= int(fmd_batch::kOscH);
  }
  // (the inputs of the two low-pass filters by call parity too, history rows in front: the decimator /
  // resampler of the next call does not wait for this call's low-pass)
  bad |= b->rdsraw[0].alloc(size_t(T_lpf - 1 + b->Rmax) * CP);
  bad |= b->rdsraw[1].alloc(size_t(T_lpf - 1 + b->Rmax) * CP);
  bad |= b->rlpf[0].alloc(size_t(b->Rmax) * CP);
  bad |= b->rlpf[1].alloc(size_t(b->Rmax) * CP);
  bad |= b->rpll.alloc(size_t(T_mf - 1 + b->Rmax) * CP);
  bad |= b->rmf.alloc(size_t(b->Rmax) * CP);
  bad |= b->tap_sync.alloc(size_t(b->Rmax) * CP);
  bad |= b->rs[0].alloc(size_t(T_alp - 1 + b->Amax) * CP);
  bad |= b->rs[1].alloc(size_t(T_alp - 1 + b->Amax) * CP);
  bad |= b->alp[0].alloc(size_t(b->Amax) * CP);
  bad |= b->alp[1].alloc(size_t(b->Amax) * CP);
  bad |= b->rds_lpf_taps.alloc(T_lpf);
  bad |= b->mf_taps2.alloc(size_t(2) * T_mf);
  bad |= b->audio_taps.alloc(2 * size_t(T_alp)); // twice in a row: k_audio_lpf_tail29 reads T taps from any a0
  b->rs_margin = fmd::rs_table_margin(d.rs_step); // zeros around every output's taps: k_resample
  b->rs_row = d.rs_order + 1 + 2 * b->rs_margin;
  bad |= b->ktab.alloc(size_t(b->Amax) * b->rs_row + 64);
  bad |= b->pidx.alloc(b->Amax);
  { // k_resample_ring
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      b->n_cus = prop.multiProcessorCount;
    size_t tab_floats = 0, head_ints = 0, step_ints = 0;
    for (int form = 0; form < 3; form++)
      if (rsr_configure(b.get(), form))
      {
        const size_t per_step = size_t(b->rsr_NW) * b->rsr_R;
        const size_t nsteps = (size_t(b->Amax) + per_step - 1) / per_step;
        tab_floats = std::max(tab_floats, nsteps * b->rsr_NW * b->rsr_nbm * 8 * b->rsr_R + 256);
        head_ints = std::max(head_ints, nsteps * b->rsr_NW * fmd::RSR_HEAD);
        step_ints = std::max(step_ints, nsteps * 2);
      }
    for (int form = 0; form < 3 && !rsr_configure(b.get(), form); form++) // the first form that fits stays
      ;
    if (tab_floats)
    {
      bad |= b->rsr_tab.alloc(tab_floats);
      bad |= b->rsr_head.alloc(head_ints);
      bad |= b->rsr_steps.alloc(step_ints);
      const void* fns[] = {reinterpret_cast<const void*>(&fmd::k_resample_ring<4, 4>),
                           reinterpret_cast<const void*>(&fmd::k_resample_ring<2, 8>),
                           reinterpret_cast<const void*>(&fmd::k_resample_ring<2, 8, true>),
                           reinterpret_cast<const void*>(&fmd::k_resample_ring<2, 4>)};
      for (const void* f : fns)
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    }
  }
  if (b->params.fir_reduction == 2)
  { // the fused multiply-add form of the resamplers is the ring kernel's, two outputs x eight waves: every call takes it
    if (b->rsr_R != 2 || b->rsr_NW != 8)
      return fail(FMD_ERR_ARG, "FMD_FIR_FMA_PARITY_WAIVED: the fused multiply-add form exists for the reference "
                               "geometry only (this resampler window does not fit the 2 x 8 ring form)");
    b->rsr_mode = 1;
  }
  bad |= b->sctab.alloc(d.sincos_tab.size());
  bad |= b->sctab256.alloc(d.sincos_tab256.size());
  bad |= b->fstate.alloc(size_t(fmd::F_SLOTS) * CP);
  bad |= b->istate.alloc(size_t(fmd::I_SLOTS) * CP);
  bad |= b->r_data.alloc(size_t(4) * CP);
  b->queue_cap = std::max(4096u, 8u * C);
  for (int q = 0; q < fmd_batch::NSLOT; q++)
    bad |= b->queue[q].alloc(b->queue_cap);
  bad |= b->queue_counts.alloc(fmd_batch::NSLOT);
  bad |= hipHostMalloc(reinterpret_cast<void**>(&b->h_counts), fmd_batch::NSLOT * sizeof(unsigned),
                       hipHostMallocDefault) != hipSuccess;
  bad |= b->export_cursor.alloc(fmd_batch::kExportCursors);
  if (bad)
    return fail(FMD_ERR_DEVICE, std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError()));

  bad |= upload(b->lut.p, lut.data(), lut.size() * sizeof(float));
  bad |= upload(b->if_coeff.p, d.if_coeff.data(), d.if_coeff.size() * sizeof(float));
  bad |= upload(b->rs_coeff.p, d.rs_coeff.data(), d.rs_coeff.size() * sizeof(float));
  bad |= upload(b->rds_lpf_taps.p, d.rds_lpf_taps.data(), T_lpf * sizeof(float));
  bad |= upload(b->audio_taps.p, d.lpf_taps.data(), T_alp * sizeof(float));
  bad |= upload(b->audio_taps.p + T_alp, d.lpf_taps.data(), T_alp * sizeof(float));
  bad |= upload(b->sctab.p, d.sincos_tab.data(), d.sincos_tab.size() * sizeof(double));
  bad |= upload(b->sctab256.p, d.sincos_tab256.data(), d.sincos_tab256.size() * sizeof(double));
  bad |= upload(b->mf_taps2.p, d.rds_mf_taps.data(), T_mf * sizeof(float));
  for (const auto& h : d.hb)
  {
    fmd::HbCoef hc{};
    for (int i = 0; i < h.len; i++)
      hc.c[i] = h.coef[size_t(i)];
    for (int j = 0; 2 * j < h.len && j < 28; j++)
      hc.e[j] = hc.c[2 * j];
    b->hbcoef.push_back(hc);
  }
  // coherent (fine-grained) host memory: the kernels' system-scope writes are visible to the host
  // without a synchronisation, whatever HIP_HOST_COHERENT says
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_err), 2 * sizeof(unsigned),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
    return fail(FMD_ERR_DEVICE, "host-mapped error word allocation failed");
  b->h_err[0] = b->h_err[1] = 0u;
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_status), size_t(fmd::HS_WORDS) * CP * sizeof(unsigned),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
    return fail(FMD_ERR_DEVICE, "host-mapped status snapshot allocation failed");
  std::fill_n(b->h_status, size_t(fmd::HS_WORDS) * CP, 0u); // a fresh decoder: all meters zero
  if (b->d_status.alloc(size_t(fmd::HS_WORDS) * CP))
    return fail(FMD_ERR_DEVICE, "status record allocation failed");
  bind_state(b.get());
  if (!b->st.err || !b->st.hs)
    return fail(FMD_ERR_DEVICE, "host-mapped memory has no device address");
  bad |= init_signal_state(b.get());
  if (bad)
    return fail(FMD_ERR_DEVICE, "upload of constants failed");
  if (do_reset(b.get())) // the cFmDecoder ctor ends with Reset() (FmDecode.cpp:313)
    return fail(FMD_ERR_DEVICE, "state reset failed");

  b->gdec.resize(C);
  if (shell)
  {
    b->owns_streams = false;
    b->s_fir = shell->s_fir;
    b->s_ser = shell->s_ser;
    b->s_post = shell->s_post;
    b->s_rds = shell->s_rds;
    b->s_lpf = shell->s_lpf;
    b->streams_sharing = shell->streams_sharing;
  }
  else if (int rc = create_streams(b.get()))
    return rc;
  // RDS chain and audio chain behind the serial stage are independent, but side by side their kernels only
  // stretch the latency-bound serial stage (-4 % at 8192 channels), so they share one stream -- except in a batch
  // of one or two wavefronts (the single decoder of cFmDecoder: RadioReceiver.cpp:515-538), which is pure
  // latency: its RDS chain (0.35 ms of lane-serial kernels) and its audio chain (0.19 ms) run side by side,
  // 2.17 -> 1.9 ms per call of one channel ("split_post" of fmd_batch_debug_set overrides).
  b->split_post = b->CP <= 128;
  // The serial stage takes whole CUs (one role wave per SIMD; at most 64 CUs = a quarter of the chip at
  // kSubBatchChannels; +4.4 % there) where the batch is big enough for the bandwidth kernels to notice their
  // neighbours at all ("serial_exclusive" of fmd_batch_debug_set overrides).
  b->serial_exclusive = b->CP <= kSubBatchChannels && b->CP >= 1024;
  for (auto& row : b->cev)
    for (auto& e : row)
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : b->ev_drained)
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  b->cev_ready = true;
  HIPCHK(hipDeviceSynchronize());
  { // can the IF stage of this geometry be launched at all (window in LDS)?  Decided here, once: a
    // process call is then never refused half-way for it
    b->if_dry_run = true;
    const std::function<void(int)> nomark = [](int) {};
    const int rc = launch_if_stage<fmd::InF32>(b.get(), nullptr, 0, FMD_MAX_BLOCK, 0, b->Mmax - 1, 0, nullptr, nomark,
                                              nullptr, nullptr);
    b->if_dry_run = false;
    if (rc != FMD_OK)
      return rc;
  }
  if (b->streams_sharing > 0 && !shell)
    // not an error: the batch works, its chains just queue behind each other where they were meant to overlap.
    // The text stays in fmd_last_error() (the call still returns FMD_OK); fmd_batch_streams_sharing_queue() has
    // the number.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the HOST sets the
    // variable before the runtime starts -- the library reads no environment variable itself).
    (void)fail(FMD_OK, std::to_string(b->streams_sharing) +
                           " of the batch's 5 internal streams share a hardware queue with another stream of this "
                           "process: overlapped calls will be slower than measured (host: GPU_MAX_HW_QUEUES=8 "
                           "before the HIP runtime initialises)");
  *out = b.release();
  return FMD_OK;
}

} // namespace

extern "C" {

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
  // (and no more channels than keep a sub-batch's row buffers below 4 GB -- create_one; only low decimations)
  unsigned sub_cap = kSubBatchChannels;
  {
    const size_t Dd = std::max(1u, params->downsample);
    const size_t mstride = ((FMD_MAX_BLOCK + Dd - 1) / Dd + 1 + 15) & ~size_t(15);
    const size_t fit = ((size_t(1) << 32) - 8192) / (mstride * sizeof(float2)) / 128 * 128;
    sub_cap = unsigned(std::max<size_t>(128, std::min<size_t>(sub_cap, fit)));
  }
  if (n_channels <= sub_cap)
    return create_one(params, n_channels, tuning_shifts, device, cb, user, nullptr, out);

  /* A shell over sub-batches (see fmd_batch::subs): equal sizes, multiples of 128 channels (a CU's worth of
   * serial stage), so that every sub-batch call is the same load. */
  std::unique_ptr<fmd_batch> b(new fmd_batch);
  b->device = device;
  if (int rc = create_streams(b.get()))
    return rc;
  const unsigned S = (n_channels + sub_cap - 1) / sub_cap;
  const unsigned per = ((n_channels + S - 1) / S + 127u) & ~127u;
  for (unsigned ch0 = 0; ch0 < n_channels; ch0 += per)
  {
    fmd_batch* sub = nullptr;
    const unsigned n = std::min(per, n_channels - ch0);
    // (callbacks: the shell runs the group decoders, with the caller's channel numbers)
    if (int rc = create_one(params, n, tuning_shifts ? tuning_shifts + ch0 : nullptr, device, nullptr, nullptr,
                            b.get(), &sub))
      return rc;
    sub->concurrency = 2; // the sub-batch calls of one call overlap; the shell orders the caller's stream (mode 1)
    b->subs.emplace_back(sub);
    b->sub_ch0.push_back(ch0);
  }
  b->sub_ch0.push_back(n_channels);
  const fmd_batch* s0 = b->subs[0].get();
  b->params = s0->params;
  b->des = s0->des;
  b->min_samples = s0->min_samples;
  b->n_cus = s0->n_cus;
  b->Mmax = s0->Mmax;
  b->Mstride = s0->Mstride;
  b->Amax = s0->Amax;
  b->C = n_channels;
  b->CP = (n_channels + 63u) & ~63u;
  for (const auto& sb : b->subs)
    b->shifts.insert(b->shifts.end(), sb->shifts.begin(), sb->shifts.end());
  if (cb)
    b->cb = *cb;
  b->user = user;
  b->gdec.resize(n_channels);
  if (b->export_cursor.alloc(fmd_batch::kExportCursors))
    return fail(FMD_ERR_DEVICE, "device allocation failed");
  HIPCHK(hipDeviceSynchronize());
  if (b->streams_sharing > 0)
    (void)fail(FMD_OK, std::to_string(b->streams_sharing) +
                           " of the batch's 5 internal streams share a hardware queue with another stream of this "
                           "process: overlapped calls will be slower than measured (host: GPU_MAX_HW_QUEUES=8 "
                           "before the HIP runtime initialises)");
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

int fmd_batch_streams_sharing_queue(const fmd_batch* b)
{
  return b ? b->streams_sharing : -1;
}

unsigned fmd_batch_min_samples(const fmd_batch* b)
{
  return b ? b->min_samples : 0;
}

unsigned fmd_batch_max_audio_floats(const fmd_batch* b, unsigned samples)
{
  if (!b)
    return 0;
  const unsigned M = (samples + b->des.D - 1) / b->des.D + 1;
  return 2 * (unsigned(double(M) / double(b->des.rs_step)) + 4);
}

} // extern "C"

#include "fmd_batch_if.inc.hpp"      // launch_if_stage: the IF stage's kernel forms
#include "fmd_batch_process.inc.hpp" // process_device_impl: one call on the batch's streams

extern "C" {

static int wait_impl(fmd_batch* b, int lag, void* stream_, bool take_lost);

/* One call of any batch: a plain one directly; a shell's as one call of every sub-batch, in channel order, on the
 * same streams (fmd_batch::subs). */
static int process_any(fmd_batch* b, const void* d_iq, IqFormat fmt, size_t iq_channel_stride, unsigned samples,
                       float* d_audio, size_t audio_channel_stride, unsigned* out_floats, void* stream)
{
  if (!b || b->subs.empty())
    return process_device_impl(b, d_iq, fmt, iq_channel_stride, samples, d_audio, audio_channel_stride, out_floats,
                               stream);
  if (!d_iq || !d_audio)
    return fail(FMD_ERR_ARG, "fmd_batch_process_device: null argument");
  if (int rc = check_device_errors(b))
    return rc;
  const size_t esz = fmt == IQ_U8 ? 2 : 8; // bytes per IQ sample
  unsigned nf = 0;
  for (size_t k = 0; k < b->subs.size(); k++)
  {
    fmd_batch* sb = b->subs[k].get();
    const unsigned ch0 = b->sub_ch0[k];
    sb->sched_prev2 = b->vheavy[1];
    const char* iq = static_cast<const char*>(d_iq) + size_t(ch0 / b->cpc) * iq_channel_stride * esz;
    const int rc = process_device_impl(sb, iq, fmt, iq_channel_stride, samples,
                                       d_audio + size_t(ch0) * audio_channel_stride, audio_channel_stride, &nf, stream);
    if (rc != FMD_OK)
    { // the first sub-batch refuses what every one of them would refuse (same geometry, same positions): nothing
      // has been submitted.  Later: part of the call is on the device -- the batch is unusable until reset.
      if (k > 0)
      {
        b->failed = true;
        b->fail_msg = std::string("a call broke off between two sub-batches: ") + g_err;
      }
      return rc;
    }
    b->vheavy[1] = b->vheavy[0];
    b->vheavy[0] = sb->cev[sb->call_index % fmd_batch::NSLOT][fmd_batch::EV_HEAVY];
  }
  b->call_index = b->subs[0]->call_index;
  b->lastM = b->subs[0]->lastM;
  b->lastA = b->subs[0]->lastA;
  b->lastR = b->subs[0]->lastR;
  if (out_floats)
    *out_floats = nf;
  if (b->concurrency == 1) // the caller's stream is ordered after every call (the sub-batches run in mode 2)
    return wait_impl(b, 0, stream, false);
  return FMD_OK;
}

int fmd_batch_process_device(fmd_batch* b, const float* d_iq, size_t iq_channel_stride,
                             unsigned samples, float* d_audio, size_t audio_channel_stride,
                             unsigned* out_floats, void* stream)
{
  return process_any(b, d_iq, IQ_F32, iq_channel_stride, samples, d_audio, audio_channel_stride, out_floats, stream);
}

int fmd_batch_process_device_u8(fmd_batch* b, const uint8_t* d_iq_u8, size_t iq_channel_stride,
                                unsigned samples, float* d_audio, size_t audio_channel_stride,
                                unsigned* out_floats, void* stream)
{
  return process_any(b, d_iq_u8, IQ_U8, iq_channel_stride, samples, d_audio, audio_channel_stride, out_floats,
                     stream);
}

/* slots whose call is at least `lag` calls old (lag 0 = every call submitted so far) */
static bool slot_eligible(const fmd_batch* b, int q, int lag)
{
  return b->slot_call[q] != 0 && b->slot_call[q] + uint32_t(lag) <= b->call_index;
}

/* take_lost = false: the recoverable flag (groups lost) stays for whoever reports it */
static int wait_impl(fmd_batch* b, int lag, void* stream_, bool take_lost)
{
  if (!b || lag < 0 || lag > 4)
    return fail(FMD_ERR_ARG, "fmd_batch_wait: bad argument (lag must be 0..4)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));
  if (is_shell(b))
  {
    for (auto& sb : b->subs)
    {
      const int rc = wait_impl(sb.get(), lag, stream_, false);
      if (rc < 0)
        return check_device_errors(b); // (marks the shell failed too)
    }
    if (int rc = check_device_errors(b))
      return rc;
    return take_lost ? take_lost_groups(b) : FMD_OK;
  }
  /* Only events that have not completed yet become waits of the caller's stream: a completed event orders nothing.
   * (Until round 6 every wait went through all eligible slots -- up to 32 barrier packets per step in the caller's
   * queue, nearly all for calls long complete.  On the null stream that cost nothing measurable; on a stream of the
   * caller's own the command processor's work on them cost 4-7 % of the whole path: tools/node_bench 275 000 against
   * 288 000 MS/s on the null stream, bench.py --side-stream 264 000 against 284 000; docs/MEASUREMENTS.md, round 6.) */
  auto wait_if_pending = [&](hipEvent_t e) {
    if (hipEventQuery(e) == hipSuccess)
      return hipSuccess;
    (void)hipGetLastError(); // (hipErrorNotReady is not an error)
    return hipStreamWaitEvent(stream, e, 0);
  };
  for (int q = 0; q < fmd_batch::NSLOT; q++)
    if (slot_eligible(b, q, lag))
    {
      HIPCHK(wait_if_pending(b->cev[q][fmd_batch::EV_AUD]));
      HIPCHK(wait_if_pending(b->cev[q][fmd_batch::EV_RDS]));
      HIPCHK(wait_if_pending(b->cev[q][fmd_batch::EV_INDONE]));
      // the history rolls behind the heavy part (br, mix, half-band tails) belong to the call too
      HIPCHK(wait_if_pending(b->cev[q][fmd_batch::EV_ROLL]));
    }
  // asynchronous: reports what the device has flagged so far (calls that have finished)
  if (int rc = check_device_errors(b))
    return rc;
  return take_lost ? take_lost_groups(b) : FMD_OK;
}

int fmd_batch_wait_lagged(fmd_batch* b, int lag, void* stream_)
{
  return wait_impl(b, lag, stream_, true);
}

int fmd_batch_take_rds_lost(fmd_batch* b)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  return take_lost_groups(b) == FMD_WARN_RDS_LOST ? 1 : 0;
}

int fmd_batch_debug_set_spin_limit(fmd_batch* b, unsigned limit)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  b->spin_limit = limit;
  b->st.spin_limit = limit;
  for (auto& sb : b->subs)
    fmd_batch_debug_set_spin_limit(sb.get(), limit);
  return FMD_OK;
}

int fmd_batch_debug_host_ms(fmd_batch* b, float out[4])
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "null argument");
  const unsigned n = b->host_calls;
  for (int i = 0; i < 4; i++)
  {
    out[i] = n ? float(b->host_ms[i] / n) : 0.0f;
    b->host_ms[i] = 0.0;
  }
  b->host_calls = 0;
  return int(n);
}

int fmd_batch_debug_set(fmd_batch* b, const char* key, int value)
{
  if (!b || !key)
    return fail(FMD_ERR_ARG, "null argument");
  const std::string k(key);
  // Keys move work between streams and change kernel forms: nothing of an earlier call may still be
  // running when the next call takes the new route -- drain the device first.
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  if (is_shell(b))
  { // every sub-batch takes the key
    for (auto& sb : b->subs)
      if (int rc = fmd_batch_debug_set(sb.get(), key, value))
        return rc;
    return FMD_OK;
  }
  if (k == "resampler")
  {
    if (value > 0 && b->rsr_R == 0)
      return fail(FMD_ERR_ARG, "fmd_batch_debug_set: no form of k_resample_ring fits this geometry");
    b->rsr_mode = value < 0 ? -1 : (value ? 1 : 0);
  }
  else if (k == "fir_nt")
    b->dbg_fir_nt = std::max(0, value);
  else if (k == "serial_claim")
    b->dbg_serial_claim = value != 0;
  else if (k == "serial_exclusive")
    b->serial_exclusive = value != 0;
  else if (k == "split_post")
    b->split_post = value != 0;
  else if (k == "hb4")
    b->dbg_hb4 = value != 0;
  else if (k == "ring4")
    b->dbg_ring4 = value != 0;
  else if (k == "lpf_late")
    b->dbg_lpf_late = value < 0 ? -1 : std::min(value, 2);
  else if (k == "stage_mask")
    b->dbg_stage_mask = value & 63;
  else if (k == "fir_ro")
    b->dbg_fir_ro = (value == 2 || value == 3) ? value : 1;
  else if (k == "serial_probe")
  { // per-workgroup timing of the serial stage's last 8 launches (fmd_batch_debug_serial_probe)
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceSynchronize());
    b->serial_probe.release();
    if (value && b->serial_probe.alloc(size_t(8) * 3 * (b->CP / 64)))
      return fail(FMD_ERR_DEVICE, "serial probe allocation failed");
  }
  else if (k == "halfband_chain") // -1 the library decides, 0 a launch per stage, 1 k_halfband_chain wherever it applies
  {
    b->hbf_mode = value < 0 ? -1 : (value ? 1 : 0);
    // before the first call a small batch can still start the batch-wide oscillator sequence (and with it
    // the form of the serial stage that writes no mixed rows); later its chain reads mixed rows
    if (value > 0 && b->call_index == 0 && b->h_osc && b->des.hb.size() == 3 &&
        b->des.hb[0].len - 1 <= int(fmd_batch::kOscH))
      b->osc_on = true;
  }
  else if (k == "nomix") // 0: the serial stage writes the mixed rows also where k_halfband_chain follows
    b->dbg_nomix = value != 0;
  else if (k == "rsr_form") // 0: 2 outputs x 8 waves, 1: 4 x 4, 2: 2 x 4 (falls through to the next that fits)
  {
    bool ok = false;
    for (int f = std::max(0, std::min(2, value)); f < 3 && !ok; f++)
      ok = rsr_configure(b, f);
  }
  else
    return fail(FMD_ERR_ARG, "fmd_batch_debug_set: unknown key '" + k + "'");
  return FMD_OK;
}

int fmd_batch_wait(fmd_batch* b, void* stream_)
{
  return fmd_batch_wait_lagged(b, 0, stream_);
}

int fmd_batch_set_channels_per_capture(fmd_batch* b, unsigned channels_per_capture)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  const unsigned k = channels_per_capture ? channels_per_capture : 1u;
  if (b->C % k)
    return fail(FMD_ERR_ARG, "fmd_batch_set_channels_per_capture: the channel count is not a multiple of it");
  for (size_t i = 0; i < b->subs.size(); i++) // a capture's channels must not straddle two sub-batches
    if (b->sub_ch0[i] % k || b->subs[i]->C % k)
      return fail(FMD_ERR_ARG, "fmd_batch_set_channels_per_capture: with " + std::to_string(b->C) +
                                   " channels the batch runs as sub-batches of " + std::to_string(b->subs[0]->C) +
                                   ", which is not a multiple of it");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  b->cpc = k;
  for (auto& sb : b->subs)
    sb->cpc = k;
  return FMD_OK;
}

int fmd_batch_set_concurrency(fmd_batch* b, int mode)
{
  if (!b || mode < 0 || mode > 2)
    return fail(FMD_ERR_ARG, "fmd_batch_set_concurrency: mode must be 0, 1 or 2");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  b->concurrency = mode;
  for (auto& sb : b->subs) // (their calls overlap inside one call of the shell; mode 1 is the shell's to keep)
    sb->concurrency = mode == 0 ? 0 : 2;
  return FMD_OK;
}

/* slots whose call is at least `lag` calls old and whose groups have not been taken out yet */
static bool slot_to_drain(const fmd_batch* b, int q, int lag)
{
  return slot_eligible(b, q, lag) && b->drained_call[q] != b->slot_call[q];
}

/* Draining the RDS queues of one batch with buffers of its own, in the three steps between which the caller
 * synchronises the stream once for ALL the batches it drains (a shell: its sub-batches). */
struct QueueDrain
{
  fmd_batch* b = nullptr;
  unsigned ch0 = 0; // what the caller adds to the records' channel numbers
  int todo[fmd_batch::NSLOT] = {}, ntodo = 0;
  unsigned cnt[fmd_batch::NSLOT] = {};
  size_t total = 0;
};

static int drain_counts(QueueDrain& d, int lag, hipStream_t stream)
{
  fmd_batch* b = d.b;
  for (int q = 0; q < fmd_batch::NSLOT; q++)
    if (slot_to_drain(b, q, lag)) // never used / already drained / its call may still be appending
    {
      HIPCHK(hipStreamWaitEvent(stream, b->cev[q][fmd_batch::EV_RDS], 0));
      d.todo[d.ntodo++] = q;
    }
  if (d.ntodo)
    HIPCHK(hipMemcpyAsync(b->h_counts, b->queue_counts.p, fmd_batch::NSLOT * sizeof(unsigned), hipMemcpyDeviceToHost,
                          stream));
  return FMD_OK;
}

static int drain_records(QueueDrain& d, hipStream_t stream)
{ // (the counts have arrived)
  fmd_batch* b = d.b;
  for (int i = 0; i < d.ntodo; i++)
  {
    d.cnt[i] = std::min(b->h_counts[d.todo[i]], b->queue_cap); // overflow: the oldest queue_cap groups are kept
    d.total += d.cnt[i];
  }
  if (d.total > b->h_recs_cap)
  {
    if (b->h_recs)
      (void)hipHostFree(b->h_recs);
    b->h_recs = nullptr;
    b->h_recs_cap = 0;
    const size_t want = std::max<size_t>(d.total, size_t(2) * b->C + 1024);
    if (hipHostMalloc(reinterpret_cast<void**>(&b->h_recs), want * sizeof(fmd::RdsGroupRec), hipHostMallocDefault) !=
        hipSuccess)
      return fail(FMD_ERR_DEVICE, "fmd_batch_collect_rds: page-locked staging allocation failed");
    b->h_recs_cap = want;
  }
  size_t at = 0;
  for (int i = 0; i < d.ntodo; i++)
    if (d.cnt[i])
    {
      const int q = d.todo[i];
      HIPCHK(hipMemcpyAsync(b->h_recs + at, b->queue[q].p, size_t(d.cnt[i]) * sizeof(fmd::RdsGroupRec),
                            hipMemcpyDeviceToHost, stream));
      HIPCHK(hipMemsetAsync(b->qcount(q), 0, sizeof(unsigned), stream));
      at += d.cnt[i];
    }
  return FMD_OK;
}

int fmd_batch_collect_rds_lagged(fmd_batch* b, fmd_rds_group* out, unsigned cap, int run_group_decoder,
                                 int lag, void* stream_)
{
  if (!b || lag < 0 || lag > 4)
    return fail(FMD_ERR_ARG, "fmd_batch_collect_rds: bad argument (lag must be 0..4)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));
  /* Two synchronisations whatever the number of queues (and of sub-batches): all counts in one copy per batch,
   * then the records of the non-empty queues back to back.  Page-locked destinations: the copies are DMA
   * transfers, not staging kernels that would queue up behind the decoder's own. */
  std::vector<QueueDrain> drains(is_shell(b) ? b->subs.size() : 1);
  for (size_t k = 0; k < drains.size(); k++)
  {
    drains[k].b = is_shell(b) ? b->subs[k].get() : b;
    drains[k].ch0 = is_shell(b) ? b->sub_ch0[k] : 0u;
  }
  bool any = false, any_recs = false;
  for (auto& d : drains)
  {
    if (int rc = drain_counts(d, lag, stream))
      return rc;
    any = any || d.ntodo > 0;
  }
  std::vector<fmd::RdsGroupRec> recs;
  if (any)
  {
    HIPCHK(hipStreamSynchronize(stream));
    for (auto& d : drains)
    {
      if (int rc = drain_records(d, stream))
        return rc;
      any_recs = any_recs || d.total > 0;
    }
    if (any_recs)
      HIPCHK(hipStreamSynchronize(stream));
    for (auto& d : drains)
    {
      for (int i = 0; i < d.ntodo; i++)
        d.b->drained_call[d.todo[i]] = d.b->slot_call[d.todo[i]];
      const size_t at = recs.size();
      recs.insert(recs.end(), d.b->h_recs, d.b->h_recs + d.total);
      for (size_t i = at; i < recs.size(); i++)
        recs[i].channel += d.ch0;
    }
  }
  if (int rc = check_device_errors(b)) // the stream was synchronised above: covers the drained calls
    return rc;
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
      for (int i = 0; i < 4; i++)
        out[k].blocks[i] = r.blocks[i];
      k++;
    }
  }
  return int(out ? k : recs.size());
}

int fmd_batch_collect_rds(fmd_batch* b, fmd_rds_group* out, unsigned cap, int run_group_decoder,
                          void* stream_)
{
  return fmd_batch_collect_rds_lagged(b, out, cap, run_group_decoder, 0, stream_);
}

/* The queues of one batch with buffers of its own, appended to the caller's record buffer at *cursor. */
static int export_queues(fmd_batch* b, int32_t* d_records, unsigned cap, unsigned channel_offset, int lag,
                         hipStream_t stream, unsigned* cursor)
{
  for (int q = 0; q < fmd_batch::NSLOT; q++)
  {
    if (!slot_to_drain(b, q, lag))
      continue; // also: already exported for its call -- one launch per call, not one per slot
    HIPCHK(hipStreamWaitEvent(stream, b->cev[q][fmd_batch::EV_RDS], 0));
    hipLaunchKernelGGL(fmd::k_rds_export, dim3(1), dim3(256), 0, stream, b->queue[q].p, b->qcount(q),
                       b->queue_cap, reinterpret_cast<int4*>(d_records), cap, cursor, channel_offset, b->st.err);
    HIPCHK(hipEventRecord(b->ev_drained[q], stream));
    b->drained_pending[q] = true;
    b->drained_call[q] = b->slot_call[q];
  }
  return FMD_OK;
}

int fmd_batch_export_rds_device(fmd_batch* b, int32_t* d_records, unsigned cap, unsigned channel_offset,
                                int lag, void* stream_)
{
  if (!b || !d_records || cap == 0 || lag < 0 || lag > 4)
    return fail(FMD_ERR_ARG, "fmd_batch_export_rds_device: bad argument (lag must be 0..4)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemsetAsync(d_records, 0, size_t(cap) * 4 * sizeof(int32_t), stream));
  unsigned* const cursor = b->export_cursor.p + (b->export_seq++ % fmd_batch::kExportCursors);
  HIPCHK(hipMemsetAsync(cursor, 0, sizeof(unsigned), stream));
  if (is_shell(b))
  {
    for (size_t k = 0; k < b->subs.size(); k++)
      if (int rc = export_queues(b->subs[k].get(), d_records, cap, channel_offset + b->sub_ch0[k], lag, stream, cursor))
        return rc;
  }
  else if (int rc = export_queues(b, d_records, cap, channel_offset, lag, stream, cursor))
    return rc;
  HIPCHK(hipGetLastError());
  if (int rc = check_device_errors(b))
    return rc;
  return take_lost_groups(b);
}

static int process_host_impl(fmd_batch* b, const void* iq, IqFormat fmt, size_t iq_channel_stride,
                             unsigned samples, float* audio, size_t audio_channel_stride,
                             unsigned* out_floats)
{
  if (!b || !iq || !audio)
    return fail(FMD_ERR_ARG, "fmd_batch_process_host: null argument");
  HIPCHK(hipSetDevice(b->device));
  const unsigned C = b->C;
  const size_t esz = fmt == IQ_U8 ? 2 : 8; // bytes per IQ sample
  // device copy: one row per channel, rows padded to a whole pair of samples
  const size_t dev_row = (size_t(samples) + 1) / 2 * 2 * esz;
  const size_t dev_iq_stride = iq_channel_stride ? dev_row / esz : 0;
  const unsigned streams = iq_channel_stride ? C / b->cpc : 1u; // input rows: one per channel, per capture, or one
  const size_t iq_floats = (dev_row * streams + 3) / 4;
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
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  clk::time_point tp = clk::now();
  if (iq_channel_stride)
    HIPCHK(hipMemcpy2D(b->h_iq.p, dev_row, iq, iq_channel_stride * esz, size_t(samples) * esz, streams,
                       hipMemcpyHostToDevice));
  else
    HIPCHK(hipMemcpy(b->h_iq.p, iq, size_t(samples) * esz, hipMemcpyHostToDevice));
  b->host_ms[0] += ms_since(tp);
  tp = clk::now();
  unsigned nf = 0;
  int rc = process_any(b, b->h_iq.p, fmt, dev_iq_stride, samples, b->h_audio.p, a_stride, &nf, nullptr);
  if (rc != FMD_OK)
    return rc;
  if (C > 1 && nf > audio_channel_stride)
    return fail(FMD_ERR_ARG, "audio_channel_stride smaller than the audio produced");
  b->host_ms[1] += ms_since(tp);
  tp = clk::now();
  // Synchronous entry point in every concurrency mode: in mode 2 the null stream is not ordered after
  // the call -- order it behind the whole call before copying the audio out.
  rc = wait_impl(b, 0, nullptr, false); // the groups-lost flag is this call's to report, at its end
  if (rc < 0)
    return rc;
  HIPCHK(hipMemcpy2D(audio, (C > 1 ? audio_channel_stride : size_t(nf)) * sizeof(float), b->h_audio.p,
                     a_stride * sizeof(float), size_t(nf) * sizeof(float), C, hipMemcpyDeviceToHost));
  b->host_ms[2] += ms_since(tp);
  tp = clk::now();
  rc = fmd_batch_collect_rds(b, nullptr, 0, 1, nullptr);
  if (rc < 0)
    return rc;
  b->host_ms[3] += ms_since(tp);
  b->host_calls++;
  if (out_floats)
    *out_floats = nf;
  return take_lost_groups(b); // FMD_OK, or FMD_WARN_RDS_LOST once: audio and state are intact
}

int fmd_batch_process_host(fmd_batch* b, const float* iq, size_t iq_channel_stride, unsigned samples,
                           float* audio, size_t audio_channel_stride, unsigned* out_floats)
{
  return process_host_impl(b, iq, IQ_F32, iq_channel_stride, samples, audio, audio_channel_stride,
                           out_floats);
}

int fmd_batch_process_host_u8(fmd_batch* b, const uint8_t* iq_u8, size_t iq_channel_stride,
                              unsigned samples, float* audio, size_t audio_channel_stride,
                              unsigned* out_floats)
{
  return process_host_impl(b, iq_u8, IQ_U8, iq_channel_stride, samples, audio, audio_channel_stride,
                           out_floats);
}

/* The getters read the status snapshot the last kernel of every call leaves in host memory: no HIP
 * call, no stream operation, nothing of the batch's bookkeeping -- Kodi's status thread polls them
 * while the demux thread is inside ProcessStream (RadioReceiver.cpp:544-572 against :524). */
int fmd_batch_get_status(fmd_batch* b, unsigned channel, fmd_status* stt)
{
  if (!b || !stt || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_get_status: bad argument");
  unsigned w[fmd::HS_WORDS], lc = 0;
  const fmd_batch* ob = owner_of(b, channel, &lc); // (a shell: the sub-batch's snapshot)
  if (!host_status_read(ob, lc, w))
    return fail(FMD_ERR_DEVICE, "fmd_batch_get_status: the status snapshot is being written and never completes");
  auto f = [&](int i) {
    float v;
    memcpy(&v, &w[i], 4);
    return v;
  };
  stt->stereo_detected = int(w[fmd::HS_STEREO]);
  // FmDecode.h:146-150
  const float tuned = float(-b->shifts[channel]) * b->des.fs_if / float(int(b->des.table_size));
  stt->tuning_offset = tuned + f(fmd::HS_BB_MEAN) * b->des.freq_dev;
  stt->interface_level = f(fmd::HS_IF_LEVEL);
  stt->baseband_level = f(fmd::HS_BB_LEVEL);
  stt->pilot_level = 2 * f(fmd::HS_P_LEVEL); // FmDecode.h:75
  stt->rds_state = int(w[fmd::HS_R_STATE]);
  return FMD_OK;
}

int fmd_batch_get_audio_level(fmd_batch* b, unsigned channel, fmd_audio_level* out)
{
  if (!b || !out || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_get_audio_level: bad argument");
  unsigned w[fmd::HS_WORDS], lc = 0;
  const fmd_batch* ob = owner_of(b, channel, &lc); // (a shell: the sub-batch's snapshot)
  if (!host_status_read(ob, lc, w))
    return fail(FMD_ERR_DEVICE, "fmd_batch_get_audio_level: the status snapshot never completes");
  memcpy(&out->mean, &w[fmd::HS_AUDIO_MEAN], 4);
  memcpy(&out->rms, &w[fmd::HS_AUDIO_RMS], 4);
  memcpy(&out->level, &w[fmd::HS_AUDIO_LEVEL], 4);
  return FMD_OK;
}

int fmd_batch_status_call_index(fmd_batch* b, unsigned channel, uint32_t* call_index)
{
  if (!b || !call_index || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_status_call_index: bad argument");
  unsigned w[fmd::HS_WORDS], lc = 0;
  const fmd_batch* ob = owner_of(b, channel, &lc); // (a shell: the sub-batch's snapshot)
  if (!host_status_read(ob, lc, w))
    return fail(FMD_ERR_DEVICE, "fmd_batch_status_call_index: the status snapshot never completes");
  *call_index = (w[fmd::HS_SEQ_END] & 0x80000000u) ? 0u : w[fmd::HS_SEQ_END];
  return FMD_OK;
}

int fmd_batch_get_tap(fmd_batch* b, int tap, unsigned channel, float* out, unsigned cap_floats)
{
  if (!b || !out || channel >= b->C)
    return fail(FMD_ERR_ARG, "fmd_batch_get_tap: bad argument");
  if (is_shell(b))
  {
    unsigned lc = 0;
    fmd_batch* ob = owner_of(b, channel, &lc);
    return fmd_batch_get_tap(ob, tap, lc, out, cap_floats);
  }
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
      HIPCHK(hipMemcpy(out, b->demod[b->call_index & 1u].p + size_t(channel) * b->Mstride, size_t(b->lastM) * 8,
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
      const char* s0 = reinterpret_cast<const char*>(b->brp(int(b->call_index & 1u))) +
                       (size_t(b->des.rs_order) * CP + channel) * 8 + (tap == FMD_TAP_PILOT38 ? 4 : 0);
      if (rows)
        HIPCHK(hipMemcpy2D(out, 4, s0, CP * 8, 4, rows, hipMemcpyDeviceToHost));
      return int(rows);
    }
    case FMD_TAP_MONO_RS:
    case FMD_TAP_STEREO_RS:
      src = reinterpret_cast<const float*>(b->rs[b->call_index & 1u].p) + (tap == FMD_TAP_MONO_RS ? 1 : 0);
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
      src = b->rlpf[b->call_index & 1u].p;
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
  for (auto& sb : b->subs)
    sb->write_taps = b->write_taps;
  return FMD_OK;
}

int fmd_batch_set_profiling(fmd_batch* b, int level)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  // Level 2 moves the next call onto the caller's stream with no event waits at all, and in
  // concurrency 2 that stream was never ordered behind the calls in flight: like set_concurrency,
  // a change of execution mode starts from an idle device.
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  b->profiling = level < 0 ? 0 : (level > 2 ? 2 : level);
  b->prof_calls = 0; // restart the averaging window
  for (auto& sb : b->subs)
  {
    sb->profiling = b->profiling;
    sb->prof_calls = 0;
  }
  return FMD_OK;
}

int fmd_batch_get_stage_ms(fmd_batch* b, float* out, unsigned cap)
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "bad argument");
  if (is_shell(b))
  { // per call of the shell: the sub-batch calls' times added up (one launch of every stage per sub-batch)
    std::vector<float> sum(ST_COUNT, 0.0f), one(ST_COUNT);
    int calls = 0;
    for (auto& sb : b->subs)
    {
      const int n = fmd_batch_get_stage_ms(sb.get(), one.data(), ST_COUNT);
      if (n < 0)
        return n;
      calls = n;
      for (int i = 0; i < ST_COUNT; i++)
        sum[size_t(i)] = (one[size_t(i)] < 0 || sum[size_t(i)] < 0) ? -1.0f : sum[size_t(i)] + one[size_t(i)];
    }
    for (unsigned i = 0; i < cap && i < unsigned(ST_COUNT); i++)
      out[i] = sum[i];
    return calls;
  }
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
  return rc < 0 ? rc : int(nf); // (a groups-lost warning does not touch the audio: fmd_last_error has it)
}

int fmd_process_stream_u8(fmd_decoder* d, const uint8_t* buf, unsigned samples, float* audio)
{
  if (!d)
    return fail(FMD_ERR_ARG, "null decoder");
  unsigned nf = 0;
  int rc = fmd_batch_process_host_u8(d->b, buf, 0, samples, audio, 0, &nf);
  return rc < 0 ? rc : int(nf); // (a groups-lost warning does not touch the audio: fmd_last_error has it)
}

fmd_batch* fmd_decoder_batch(fmd_decoder* d)
{
  return d ? d->b : nullptr;
}

int fmd_get_status(fmd_decoder* d, fmd_status* st)
{
  return d ? fmd_batch_get_status(d->b, 0, st) : fail(FMD_ERR_ARG, "null decoder");
}

/* Test aid, see k_debug_math.  Host arrays in and out; b may be NULL for one-argument functions. */
int fmd_debug_math(int what, unsigned n, const float* a, const float* b, float* out0, float* out1)
{
  if (!a || !out0 || !out1 || n == 0)
    return fail(FMD_ERR_ARG, "fmd_debug_math: bad argument");
  fmd::Params p{};
  p.sample_rate_if = 2.4e6;
  p.sample_rate_pcm = 48000.0;
  p.bandwidth_pcm = 15000.0;
  p.downsample = 11;
  const fmd::Design d = fmd::make_design(p);
  DevBuf<float> da, db, d0, d1;
  DevBuf<double> tab, tab256;
  int bad = da.alloc(n) | db.alloc(n) | d0.alloc(n) | d1.alloc(n) | tab.alloc(d.sincos_tab.size()) |
            tab256.alloc(d.sincos_tab256.size());
  if (!bad)
  {
    bad |= hipMemcpy(da.p, a, size_t(n) * 4, hipMemcpyHostToDevice) != hipSuccess;
    if (b)
      bad |= hipMemcpy(db.p, b, size_t(n) * 4, hipMemcpyHostToDevice) != hipSuccess;
    bad |= hipMemcpy(tab.p, d.sincos_tab.data(), d.sincos_tab.size() * 8, hipMemcpyHostToDevice) !=
           hipSuccess;
    bad |= hipMemcpy(tab256.p, d.sincos_tab256.data(), d.sincos_tab256.size() * 8, hipMemcpyHostToDevice) !=
           hipSuccess;
  }
  if (!bad)
  {
    hipLaunchKernelGGL(fmd::k_debug_math, dim3(std::min(4096u, (n + 63) / 64)), dim3(64), 0, nullptr, what,
                       n, da.p, db.p, d0.p, d1.p, tab.p,
                       FmdSincosTab{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo}, tab256.p);
    bad |= hipDeviceSynchronize() != hipSuccess;
    bad |= hipMemcpy(out0, d0.p, size_t(n) * 4, hipMemcpyDeviceToHost) != hipSuccess;
    bad |= hipMemcpy(out1, d1.p, size_t(n) * 4, hipMemcpyDeviceToHost) != hipSuccess;
  }
  da.release();
  db.release();
  d0.release();
  d1.release();
  tab.release();
  tab256.release();
  return bad ? fail(FMD_ERR_DEVICE, "fmd_debug_math: device error") : FMD_OK;
}

int fmd_batch_debug_timeline(fmd_batch* b, float* out, unsigned cap_calls)
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "fmd_batch_debug_timeline: null argument");
  if (is_shell(b)) // (the first sub-batch's calls: every S-th of the common sequence)
    return fmd_batch_debug_timeline(b->subs[0].get(), out, cap_calls);
  if (b->profiling != 1 || b->prof_calls == 0 || !b->serial_exclusive || b->concurrency != 2)
    return 0;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  const unsigned n = std::min(cap_calls, b->prof_calls);
  hipEvent_t t0 = b->ev[0];
  for (unsigned c = 0; c < n; c++)
  {
    hipEvent_t* es = &b->ev[size_t(c) * (ST_COUNT + 1)];
    for (int i = 0; i < 10; i++)
    {
      float ms = -1.0f;
      if (hipEventElapsedTime(&ms, t0, es[i]) != hipSuccess)
        ms = -1.0f; // an event that was never recorded (the call's tail was not profiled)
      out[size_t(c) * 10 + i] = ms;
    }
  }
  (void)hipGetLastError();
  return int(n);
}

/* Which of the batch's internal streams share a hardware queue with `stream` (the caller's): bit i = internal
 * stream i (0 IF FIR, 1 serial stage, 2 heavy, 3 light RDS half, 4 light audio half / filters) had to wait for a wave
 * that kept `stream` busy; bit 8 + i = `stream` had to wait for internal stream i.  Drains the device. */
int fmd_batch_debug_stream_conflicts(fmd_batch* b, void* stream_)
{
  if (!b)
    return fail(FMD_ERR_ARG, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  hipStream_t caller = static_cast<hipStream_t>(stream_);
  hipStream_t in[5] = {b->s_fir, b->s_ser, b->s_post, b->s_rds, b->s_lpf};
  const long long spin = 1200000; // ~0.5 ms
  int mask = 0;
  auto waits_behind = [&](hipStream_t busy, hipStream_t probe) {
    hipLaunchKernelGGL(fmd::k_probe_spin, dim3(1), dim3(64), 0, busy, spin, nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(fmd::k_probe_nop, dim3(1), dim3(64), 0, probe, nullptr);
    (void)hipStreamSynchronize(probe);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    (void)hipStreamSynchronize(busy);
    return us >= 250.0;
  };
  for (int i = 0; i < 5; i++)
    if (in[i])
    {
      if (waits_behind(caller, in[i]))
        mask |= 1 << i;
      if (waits_behind(in[i], caller))
        mask |= 1 << (8 + i);
    }
  return mask;
}

int fmd_batch_debug_serial_probe(fmd_batch* b, long long* out, unsigned cap_workgroups)
{
  if (!b || !out)
    return fail(FMD_ERR_ARG, "fmd_batch_debug_serial_probe: null argument");
  if (is_shell(b))
    return fmd_batch_debug_serial_probe(b->subs[0].get(), out, cap_workgroups);
  if (!b->serial_probe.p)
    return 0;
  const unsigned wgs = std::min<unsigned>(cap_workgroups, unsigned(b->serial_probe.n / 3));
  // 8 launches x (CP / 64) slots, launch = call index mod 8 (the exclusive form uses every other slot's worth)
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(out, b->serial_probe.p, size_t(wgs) * 3 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(FMD_ERR_DEVICE, "fmd_batch_debug_serial_probe: copy failed");
  return int(wgs);
}

/* ---- cRadioReceiver's stream side (csrc/fmd_receiver.hpp) ------------------------------------ */
} // extern "C"

fmd_batch* fmd::Receiver::fmd_decoder_batch(fmd_decoder* d)
{
  return d ? d->b : nullptr;
}

struct fmd_receiver
{
  fmd::Receiver r;
  fmd_receiver(const fmd_params& p, double tuner_freq, const char* name) : r(p, tuner_freq, name) {}
};

extern "C" {

int fmd_receiver_open(const fmd_params* params, double tuner_freq, const char* adapter_name,
                      fmd_receiver** out)
{
  if (!params || !out)
    return fail(FMD_ERR_ARG, "fmd_receiver_open: null argument");
  *out = nullptr;
  std::unique_ptr<fmd_receiver> r(new fmd_receiver(*params, tuner_freq, adapter_name));
  const int rc = r->r.Open(*params);
  if (rc != FMD_OK)
    return rc;
  *out = r.release();
  return FMD_OK;
}

void fmd_receiver_close(fmd_receiver* r)
{
  delete r;
}

int fmd_receiver_write_iq(fmd_receiver* r, const float* iq, unsigned samples)
{
  if (!r || (!iq && samples))
    return fail(FMD_ERR_ARG, "fmd_receiver_write_iq: null argument");
  if (!r->r.Write(iq, samples, false))
    return fail(FMD_ERR_DEVICE, "fmd_receiver_write_iq: no page-locked memory for the block");
  return FMD_OK;
}

int fmd_receiver_write_u8(fmd_receiver* r, const uint8_t* buf, unsigned samples)
{
  if (!r || (!buf && samples))
    return fail(FMD_ERR_ARG, "fmd_receiver_write_u8: null argument");
  if (!r->r.Write(buf, samples, true))
    return fail(FMD_ERR_DEVICE, "fmd_receiver_write_u8: no page-locked memory for the block");
  return FMD_OK;
}

void fmd_receiver_end(fmd_receiver* r)
{
  if (r)
    r->r.End();
}

size_t fmd_receiver_queued_samples(fmd_receiver* r)
{
  return r ? r->r.QueuedSamples() : 0;
}

void fmd_receiver_set_stream_change(fmd_receiver* r)
{
  if (r)
    r->r.SetStreamChange();
}

int fmd_receiver_demux_read(fmd_receiver* r, fmd_demux_packet* pkt)
{
  if (!r || !pkt)
    return fail(FMD_ERR_ARG, "fmd_receiver_demux_read: null argument");
  return r->r.NextPacket(pkt);
}

int fmd_receiver_signal_status(fmd_receiver* r, float* interface_level_db, float* audio_level_db,
                               int* stereo)
{
  if (!r || !interface_level_db || !audio_level_db || !stereo)
    return fail(FMD_ERR_ARG, "fmd_receiver_signal_status: null argument");
  bool st = false;
  if (!r->r.Levels(*interface_level_db, *audio_level_db, st))
    return 0;
  *stereo = st ? 1 : 0;
  return 1;
}

int fmd_receiver_pvr_signal_status(fmd_receiver* r, fmd_pvr_signal_status* out)
{
  if (!r || !out)
    return fail(FMD_ERR_ARG, "fmd_receiver_pvr_signal_status: null argument");
  return r->r.PvrStatus(*out) ? 1 : 0;
}

fmd_decoder* fmd_receiver_decoder(fmd_receiver* r)
{
  return r ? r->r.Decoder() : nullptr;
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

static int copy_design(const std::vector<float>& v, float* out, unsigned cap)
{
  if (!out && cap)
    return -1;
  for (size_t i = 0; i < v.size() && i < cap; i++)
    out[i] = v[i];
  return int(v.size());
}

int fmd_design_lanczos(unsigned order, double cutoff, float* out, unsigned cap)
{
  if (order < 1)
    return -1;
  return copy_design(fmd::make_lanczos(order, cutoff), out, cap);
}

int fmd_design_lp_kaiser(float scale, float astop, float fpass, float fstop, float fs, float* out,
                         unsigned cap)
{
  return copy_design(fmd::make_kaiser_lp(scale, astop, fpass, fstop, fs), out, cap);
}

int fmd_design_biquad(int type, float f0, float q, float fs, float out[5])
{
  if (!out || type < 0 || type > 3)
    return -1;
  const fmd::Biquad b = fmd::make_biquad(fmd::BiquadType(type), f0, q, fs);
  out[0] = b.b0;
  out[1] = b.b1;
  out[2] = b.b2;
  out[3] = b.a1;
  out[4] = b.a2;
  return 5;
}

int fmd_design_tuner_lut(unsigned table_size, int freq_shift, float* out, unsigned cap)
{
  if (!table_size)
    return -1;
  return copy_design(fmd::make_tuner_lut(table_size, freq_shift), out, cap);
}

int fmd_uecp_stuff_frame(const uint8_t* frame, unsigned len, uint8_t* out, unsigned cap)
{
  if (!frame && len)
    return -1;
  unsigned k = 0;
  fmd::uecp_stuff(frame, len, [&](uint8_t v) {
    if (k < cap)
      out[k] = v;
    k++;
  });
  return k <= cap ? int(k) : -1;
}

} // extern "C"
