/*
 * fmd_gather.hip -- rank-0 gather of float audio and RDS records over RCCL (include/fmd_gather.h).
 * Host code (its one kernel is a measurement aid: fmd_gather_debug_emulate_peers); its own library so that
 * libfmd_hip.so does not need RCCL.
 *
 * Build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared fmd_gather.hip -o libfmd_gather.so
 *        -L.. -lfmd_hip -lrccl
 */
#include "../../include/fmd_gather.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

namespace
{

thread_local std::string g_gerr;

int gfail(int code, const std::string& msg)
{
  g_gerr = msg;
  return code;
}

#define GHIP(expr)                                                                              \
  do                                                                                            \
  {                                                                                             \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return gfail(FMD_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)
#define GNCCL(expr)                                                                             \
  do                                                                                            \
  {                                                                                             \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      return gfail(FMD_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_));         \
  } while (0)

static_assert(sizeof(ncclUniqueId) == FMD_GATHER_ID_BYTES, "ncclUniqueId size");

#ifdef __HIPCC__
/* fmd_gather_debug_emulate_peers: the stores of `peers` receives -- workgroup w of peer p fills its share of that
 * peer's n16 16-byte words (audio) and, the peer's first workgroup, of its record words. */
__global__ __launch_bounds__(512) void k_emulate_recv(float4* audio, size_t audio16_per_rank, int4* rds,
                                                      size_t rds16_per_rank, unsigned wgs_per_peer, unsigned tag)
{
  const unsigned p = blockIdx.x / wgs_per_peer + 1u, w = blockIdx.x % wgs_per_peer;
  float4* a = audio + size_t(p) * audio16_per_rank;
  const float v = __uint_as_float(0x3f000000u | (tag & 0xffffu));
  for (size_t i = size_t(w) * 512 + threadIdx.x; i < audio16_per_rank; i += size_t(wgs_per_peer) * 512)
    a[i] = make_float4(v, v, v, v);
  if (w == 0)
    for (size_t i = threadIdx.x; i < rds16_per_rank; i += 512)
      rds[size_t(p) * rds16_per_rank + i] = make_int4(0, 0, 0, 0); // (no group: a zero row is padding)
}
/* ... and a sender's share: its message read once (the bytes leave over xGMI, nothing is written locally) */
__global__ __launch_bounds__(512) void k_emulate_send(const float4* audio, size_t audio16, unsigned* sink)
{
  float acc = 0.0f;
  const size_t stride = size_t(gridDim.x) * 512;
  size_t i = size_t(blockIdx.x) * 512 + threadIdx.x;
  for (; i + 3 * stride < audio16; i += 4 * stride)
  { // four loads in flight per lane: a few workgroups read at the rate a link can take
    const float4 a = audio[i], b = audio[i + stride], c = audio[i + 2 * stride], d = audio[i + 3 * stride];
    acc += a.x + b.y + c.z + d.w;
  }
  for (; i < audio16; i += stride)
    acc += audio[i].x;
  if (acc == 1.2345e38f) // never (keeps the loads)
    *sink = 1u;
}
#endif

} // namespace

struct fmd_gather
{
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  size_t audio_floats = 0;
  unsigned rds_rows = 0;
  hipStream_t side = nullptr;      // the gathers' own stream: they overlap the next steps' compute
  hipEvent_t ready = nullptr;      // caller's stream -> side
  static constexpr unsigned kRing = 16;
  hipEvent_t done[kRing] = {};     // side -> whoever reuses the buffers, one per step in flight
  uint64_t issued = 0;
  static constexpr size_t kTimed = 1024; // timing covers at most the last kTimed steps since the last query
  std::vector<hipEvent_t> t0, t1;  // timing of the steps since the last query (a ring of kTimed pairs)
  size_t timed = 0;
  double* d_word = nullptr;        // the barrier's all-reduce
  bool failed = false;             // a step's send / receive failed: the communicator is in an unknown state
  int emu_peers = 0, emu_wgs = 2;  // fmd_gather_debug_emulate_peers
  int emu_every = 1;               // fmd_gather_debug_emulate_role: receives every n-th step (0: never), sends otherwise
};

extern "C" {

const char* fmd_gather_last_error(void)
{
  return g_gerr.c_str();
}

int fmd_gather_unique_id(uint8_t id[FMD_GATHER_ID_BYTES])
{
  if (!id)
    return gfail(FMD_ERR_ARG, "null id");
  ncclUniqueId u;
  GNCCL(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return FMD_OK;
}

int fmd_gather_create(const uint8_t id[FMD_GATHER_ID_BYTES], int rank, int world, int device,
                      size_t audio_floats, unsigned rds_rows, fmd_gather** out)
{
  if (!id || !out || world < 1 || rank < 0 || rank >= world || audio_floats == 0 || rds_rows == 0)
    return gfail(FMD_ERR_ARG, "fmd_gather_create: bad argument");
  *out = nullptr;
  GHIP(hipSetDevice(device));
  fmd_gather* g = new fmd_gather;
  g->rank = rank;
  g->world = world;
  g->device = device;
  g->audio_floats = audio_floats;
  g->rds_rows = rds_rows;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = ncclCommInitRank(&g->comm, world, u, rank);
  if (r != ncclSuccess)
  {
    delete g;
    return gfail(FMD_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  if (hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&g->ready, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&g->d_word), 2 * sizeof(double)) != hipSuccess)
  {
    fmd_gather_destroy(g);
    return gfail(FMD_ERR_DEVICE, "fmd_gather_create: stream / event / buffer creation failed");
  }
  for (auto& e : g->done)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
    {
      fmd_gather_destroy(g);
      return gfail(FMD_ERR_DEVICE, "fmd_gather_create: event creation failed");
    }
  *out = g;
  return FMD_OK;
}

void fmd_gather_destroy(fmd_gather* g)
{
  if (!g)
    return;
  (void)hipSetDevice(g->device);
  if (g->side)
    (void)hipStreamSynchronize(g->side);
  if (g->comm) // (after a failed step: abort -- a destroy would wait for the half-done collective)
    (void)(g->failed ? ncclCommAbort(g->comm) : ncclCommDestroy(g->comm));
  for (auto e : g->t0)
    (void)hipEventDestroy(e);
  for (auto e : g->t1)
    (void)hipEventDestroy(e);
  if (g->ready)
    (void)hipEventDestroy(g->ready);
  for (auto e : g->done)
    if (e)
      (void)hipEventDestroy(e);
  if (g->side)
    (void)hipStreamDestroy(g->side);
  if (g->d_word)
    (void)hipFree(g->d_word);
  delete g;
}

int fmd_gather_step(fmd_gather* g, fmd_batch* batch, int lag, unsigned channel_offset, const float* d_audio,
                    int32_t* d_rds, float* d_all_audio, int32_t* d_all_rds, void* stream_)
{
  return fmd_gather_step_root(g, 0, batch, lag, channel_offset, d_audio, d_rds, d_all_audio, d_all_rds, stream_);
}

int fmd_gather_step_root(fmd_gather* g, int root, fmd_batch* batch, int lag, unsigned channel_offset,
                         const float* d_audio, int32_t* d_rds, float* d_all_audio, int32_t* d_all_rds, void* stream_)
{
  if (!g || root < 0 || root >= g->world)
    return gfail(FMD_ERR_ARG, "fmd_gather_step: null gather or root outside the world");
  if (!d_audio || !d_rds || (g->rank == root && (!d_all_audio || !d_all_rds)))
    return gfail(FMD_ERR_ARG, "fmd_gather_step: null buffer");
  if (g->failed)
    return gfail(FMD_ERR_STATE, "fmd_gather_step: an earlier step's send / receive failed -- the peers are inside a "
                                "half-done step; tear the communicator down (fmd_gather_destroy) on every rank");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GHIP(hipSetDevice(g->device));
  int warn = FMD_OK;
  if (batch)
  {
    const int rc = fmd_batch_export_rds_device(batch, d_rds, g->rds_rows, channel_offset, lag, stream);
    if (rc < 0)
      return gfail(rc, std::string("fmd_batch_export_rds_device: ") + fmd_last_error());
    warn = rc;
  }
  GHIP(hipEventRecord(g->ready, stream));
  GHIP(hipStreamWaitEvent(g->side, g->ready, 0));
  const size_t tslot = g->timed % fmd_gather::kTimed; // a caller that never asks for timings keeps 1024 pairs
  if (tslot == g->t0.size())
  {
    hipEvent_t a, b;
    GHIP(hipEventCreate(&a));
    GHIP(hipEventCreate(&b));
    g->t0.push_back(a);
    g->t1.push_back(b);
  }
  GHIP(hipEventRecord(g->t0[tslot], g->side));
  const size_t rds_ints = size_t(g->rds_rows) * 4;
  GNCCL(ncclGroupStart());
  { // a failed send / receive must not leave the group open: later calls on the communicator would hang in it
    ncclResult_t r_ = ncclSuccess;
    const char* what = "";
    auto op = [&](ncclResult_t r, const char* w) {
      if (r_ == ncclSuccess && r != ncclSuccess)
      {
        r_ = r;
        what = w;
      }
      return r_ == ncclSuccess;
    };
    if (g->rank != root)
    {
      if (op(ncclSend(d_audio, g->audio_floats, ncclFloat, root, g->comm, g->side), "ncclSend(audio)"))
        op(ncclSend(d_rds, rds_ints, ncclInt32, root, g->comm, g->side), "ncclSend(rds)");
    }
    else
      for (int r = 0; r < g->world && r_ == ncclSuccess; r++)
      {
        if (r == root)
          continue;
        if (op(ncclRecv(d_all_audio + size_t(r) * g->audio_floats, g->audio_floats, ncclFloat, r, g->comm, g->side),
               "ncclRecv(audio)"))
          op(ncclRecv(d_all_rds + size_t(r) * rds_ints, rds_ints, ncclInt32, r, g->comm, g->side), "ncclRecv(rds)");
      }
    if (r_ != ncclSuccess)
    { // the group still issues what was queued before the failure: this step is incomplete on some peer.  No later
      // step is taken (the caller tears the communicator down: fmd_gather_destroy aborts it), and the step's open
      // timing pair is dropped (t0 without t1)
      (void)ncclGroupEnd();
      g->failed = true;
      return gfail(FMD_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(r_) +
                                       " (the gather takes no further steps; destroy it on every rank)");
    }
  }
  GNCCL(ncclGroupEnd());
#ifdef __HIPCC__
  if (g->emu_peers > 0 && (g->emu_every == 0 || g->issued % uint64_t(g->emu_every) != 0))
    hipLaunchKernelGGL(k_emulate_send, dim3(unsigned(8 * g->emu_wgs)), dim3(512), 0, g->side,
                       reinterpret_cast<const float4*>(d_audio), g->audio_floats / 4,
                       reinterpret_cast<unsigned*>(g->d_word));
  else if (g->emu_peers > 0) // measurement aid: what `emu_peers` receives would write (world of one)
    hipLaunchKernelGGL(k_emulate_recv, dim3(unsigned(g->emu_peers * g->emu_wgs)), dim3(512), 0, g->side,
                       reinterpret_cast<float4*>(d_all_audio), g->audio_floats / 4, reinterpret_cast<int4*>(d_all_rds),
                       size_t(g->rds_rows), unsigned(g->emu_wgs), unsigned(g->issued));
#endif
  if (g->rank == root)
  { // the root's own outputs: a device copy, on the same stream -- unless the caller had them produced in place
    // (d_audio == its slot of d_all_audio: 88 MB per step at 8192 channels that need not be read and written again)
    float* own_a = d_all_audio + size_t(root) * g->audio_floats;
    int32_t* own_r = d_all_rds + size_t(root) * rds_ints;
    if (own_a != d_audio)
      GHIP(hipMemcpyAsync(own_a, d_audio, g->audio_floats * sizeof(float), hipMemcpyDeviceToDevice, g->side));
    if (own_r != d_rds)
      GHIP(hipMemcpyAsync(own_r, d_rds, rds_ints * sizeof(int32_t), hipMemcpyDeviceToDevice, g->side));
  }
  GHIP(hipEventRecord(g->t1[tslot], g->side));
  g->timed++;
  GHIP(hipEventRecord(g->done[g->issued % fmd_gather::kRing], g->side));
  g->issued++;
  return warn;
}

int fmd_gather_wait_lagged(fmd_gather* g, unsigned lag, void* stream_)
{
  if (!g || lag >= fmd_gather::kRing)
    return gfail(FMD_ERR_ARG, "fmd_gather_wait: null gather or lag >= 16");
  if (g->issued > lag) // the steps complete in order on the library's stream: the youngest one waited for covers the rest
    GHIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream_), g->done[(g->issued - 1 - lag) % fmd_gather::kRing], 0));
  return FMD_OK;
}

int fmd_gather_wait(fmd_gather* g, void* stream_)
{
  return fmd_gather_wait_lagged(g, 0, stream_);
}

int fmd_gather_barrier(fmd_gather* g, double value, double* max_value)
{
  if (!g)
    return gfail(FMD_ERR_ARG, "null gather");
  GHIP(hipSetDevice(g->device));
  GHIP(hipMemcpyAsync(g->d_word, &value, sizeof(double), hipMemcpyHostToDevice, g->side));
  GNCCL(ncclAllReduce(g->d_word, g->d_word + 1, 1, ncclDouble, ncclMax, g->comm, g->side));
  double m = value;
  GHIP(hipMemcpyAsync(&m, g->d_word + 1, sizeof(double), hipMemcpyDeviceToHost, g->side));
  GHIP(hipStreamSynchronize(g->side));
  if (max_value)
    *max_value = m;
  return FMD_OK;
}

float fmd_gather_ms_per_step(fmd_gather* g)
{
  if (!g || g->timed == 0)
    return -1.0f;
  if (hipStreamSynchronize(g->side) != hipSuccess)
    return -1.0f;
  double sum = 0.0;
  const size_t n = g->timed < fmd_gather::kTimed ? g->timed : fmd_gather::kTimed;
  for (size_t i = 0; i < n; i++)
  {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, g->t0[i], g->t1[i]) == hipSuccess)
      sum += ms;
  }
  const float mean = float(sum / double(n));
  g->timed = 0;
  return mean;
}

int fmd_gather_debug_emulate_peers(fmd_gather* g, int peers, int workgroups_per_peer)
{
  if (!g || peers < 0 || peers > 63 || workgroups_per_peer < 1 || workgroups_per_peer > 64)
    return gfail(FMD_ERR_ARG, "fmd_gather_debug_emulate_peers: bad argument");
  if (g->world != 1)
    return gfail(FMD_ERR_STATE, "fmd_gather_debug_emulate_peers: a world of one only (real peers send real data)");
  if (g->audio_floats % 4)
    return gfail(FMD_ERR_ARG, "fmd_gather_debug_emulate_peers: audio_floats must be a multiple of 4");
#ifndef __HIPCC__
  if (peers)
    return gfail(FMD_ERR_STATE, "fmd_gather_debug_emulate_peers: built without device code");
#endif
  g->emu_peers = peers;
  g->emu_wgs = workgroups_per_peer;
  return FMD_OK;
}

int fmd_gather_debug_emulate_role(fmd_gather* g, int every)
{
  if (!g || every < 0 || every > 64)
    return gfail(FMD_ERR_ARG, "fmd_gather_debug_emulate_role: bad argument");
  g->emu_every = every;
  return FMD_OK;
}

int fmd_gather_info(fmd_gather* g, fmd_gather_info_t* out)
{
  if (!g || !out)
    return gfail(FMD_ERR_ARG, "null argument");
  int count = 0, rank = -1, dev = -1;
  GNCCL(ncclCommCount(g->comm, &count));
  GNCCL(ncclCommUserRank(g->comm, &rank));
  GNCCL(ncclCommCuDevice(g->comm, &dev));
  out->ranks_seen = count;
  out->rank = rank;
  out->device = dev;
  out->world_asked = g->world;
  out->steps_issued = g->issued;
  return FMD_OK;
}

} // extern "C"
