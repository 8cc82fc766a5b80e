/*
 * fake_hip_rccl.cpp -- TEST DOUBLE of the HIP runtime calls and the RCCL calls that
 * pvr.rtl.radiofm_amd/csrc/fmd_gather.hip makes, so that the whole-node gather (include/fmd_gather.h) can run
 * with a world of two or three ranks in the build container, which has neither a GPU nor a peer.
 * tests/ only: nothing in the product links it (tests/test_gather_double.py builds fmd_gather.hip with g++
 * against this file instead of libamdhip64 / librccl).
 *
 * What it models honestly:
 *   - streams are in-order queues run by a worker thread each; hipEventRecord / hipStreamWaitEvent order
 *     streams the way HIP does (a wait captures the event's latest record at call time); "device" memory is
 *     host memory.  A gather that forgot to order its side stream behind the caller's would read stale data.
 *   - ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd become ONE stream operation that progresses
 *     all of the group's transfers together (rings in POSIX shared memory, one per (source, destination));
 *     a send pairs with the receive of the same order on the other side, sizes must match.
 *   - ncclAllReduce of doubles with ncclMax (the barrier), ncclCommCount / UserRank / CuDevice.
 * What it checks itself: transfers outside their communicator's world, size mismatches between the two
 * ends, a group left open at ncclCommDestroy (abort with a message).
 * Test hooks (environment, read once): FAKE_RCCL_FAIL_RECV=k makes the k-th ncclRecv of a process return
 * ncclInternalError; fake_rccl_group_depth() reports the calling thread's open groups; fake_hip_stream_busy()
 * keeps a stream busy for a while (the caller's kernels that produce what the gather sends).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace
{
[[noreturn]] void die(const char* what)
{
  fprintf(stderr, "fake_hip_rccl: %s\n", what);
  abort();
}

/* ---------------- streams and events ---------------- */
struct FakeStream
{
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  std::thread th;
  FakeStream() : th([this] { run(); }) {}
  ~FakeStream()
  {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void run()
  {
    for (;;)
    {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || !q.empty(); });
        if (q.empty())
          return;
        f = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      f();
      {
        std::lock_guard<std::mutex> l(m);
        busy = false;
      }
      cv.notify_all();
    }
  }
  void push(std::function<void()> f)
  {
    {
      std::lock_guard<std::mutex> l(m);
      q.push_back(std::move(f));
    }
    cv.notify_all();
  }
  void sync()
  {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return q.empty() && !busy; });
  }
};

struct FakeEvent
{
  std::mutex m;
  std::condition_variable cv;
  uint64_t recorded = 0, done = 0; // tickets
  std::chrono::steady_clock::time_point when;
};

FakeStream* null_stream()
{
  static FakeStream* s = new FakeStream;
  return s;
}
FakeStream* S(hipStream_t s)
{
  return s ? reinterpret_cast<FakeStream*>(s) : null_stream();
}
FakeEvent* E(hipEvent_t e)
{
  return reinterpret_cast<FakeEvent*>(e);
}

/* ---------------- the shared-memory "fabric" ---------------- */
constexpr int kMaxRanks = 8;
constexpr size_t kRingBytes = 64 * 1024; // smaller than the test's messages: transfers wrap and stall
struct Ring
{
  std::atomic<uint64_t> head, tail; // bytes written / read
  std::atomic<uint64_t> sizes[64];  // message sizes, in order (checked against the receiver's)
  std::atomic<uint64_t> nsent, nrecv;
  unsigned char data[kRingBytes];
};
struct Shared
{
  std::atomic<int> joined; // ranks that have ever joined (what ncclCommCount reports)
  std::atomic<int> alive;  // ... and have not left yet (the last one removes the segment)
  std::atomic<uint64_t> arrive[2];
  double vals[2][kMaxRanks];
  Ring ring[kMaxRanks][kMaxRanks]; // [src][dst]
};

struct Op
{
  bool send;
  void* buf;
  size_t bytes, at = 0;
  int peer;
  bool announced = false, done = false;
};
} // namespace

struct ncclComm
{
  int rank, world;
  Shared* sh;
  uint64_t ar_gen = 0;
  char name[64];
};

namespace
{
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local ncclComm* g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;
std::atomic<int> g_recv_calls{0};

int fail_recv_at()
{
  static int v = [] {
    const char* e = getenv("FAKE_RCCL_FAIL_RECV");
    return e ? atoi(e) : 0;
  }();
  return v;
}

void run_group(ncclComm* c, std::vector<Op> ops)
{
  size_t left = ops.size();
  while (left)
  {
    bool moved = false;
    for (size_t x = 0; x < ops.size(); x++)
    {
      Op& o = ops[x];
      if (o.done)
        continue;
      // transfers between the same two ranks in the same direction share a ring: strictly in order
      bool blocked = false;
      for (size_t y = 0; y < x && !blocked; y++)
        blocked = !ops[y].done && ops[y].send == o.send && ops[y].peer == o.peer;
      if (blocked)
        continue;
      Ring& r = o.send ? c->sh->ring[c->rank][o.peer] : c->sh->ring[o.peer][c->rank];
      if (!o.announced)
      {
        if (o.send)
        {
          const uint64_t k = r.nsent.load();
          if (k - r.nrecv.load() >= 64)
            continue;
          r.sizes[k % 64].store(o.bytes);
          r.nsent.store(k + 1);
        }
        else
        { // the matching send is the nrecv-th message of the ring
          const uint64_t k = r.nrecv.load();
          if (r.nsent.load() <= k)
            continue;
          if (r.sizes[k % 64].load() != o.bytes)
            die("a receive's size differs from the matching send's");
        }
        o.announced = true;
        moved = true;
      }
      const uint64_t h = r.head.load(), t = r.tail.load();
      unsigned char* buf = static_cast<unsigned char*>(o.buf);
      if (o.send)
      {
        const size_t n = std::min(kRingBytes - size_t(h - t), o.bytes - o.at);
        for (size_t i = 0; i < n; i++)
          r.data[(h + i) % kRingBytes] = buf[o.at + i];
        r.head.store(h + n);
        o.at += n;
        moved = moved || n;
      }
      else
      {
        const size_t n = std::min(size_t(h - t), o.bytes - o.at);
        for (size_t i = 0; i < n; i++)
          buf[o.at + i] = r.data[(t + i) % kRingBytes];
        r.tail.store(t + n);
        o.at += n;
        moved = moved || n;
      }
      if (o.at == o.bytes)
      {
        if (!o.send)
          r.nrecv.fetch_add(1);
        o.done = true;
        left--;
      }
    }
    if (!moved)
      std::this_thread::yield();
  }
}

size_t type_bytes(ncclDataType_t t)
{
  switch (t)
  {
    case ncclFloat:
    case ncclInt32:
    case ncclUint32:
      return 4;
    case ncclDouble:
    case ncclInt64:
    case ncclUint64:
      return 8;
    case ncclInt8:
    case ncclUint8:
      return 1;
    default:
      die("data type the double does not know");
  }
}

ncclResult_t add_op(bool send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm* c, hipStream_t s)
{
  if (!c || peer < 0 || peer >= c->world || peer == c->rank)
    return ncclInvalidArgument;
  if (g_depth && g_comm && (g_comm != c || g_stream != s))
    die("one group, two communicators or streams: the double does not model that");
  g_comm = c;
  g_stream = s;
  g_ops.push_back(Op{send, const_cast<void*>(buf), count * type_bytes(t), 0, peer});
  if (g_depth == 0)
  {
    std::vector<Op> ops;
    ops.swap(g_ops);
    S(s)->push([c, ops] { run_group(c, ops); });
  }
  return ncclSuccess;
}
} // namespace

extern "C" {

int fake_rccl_group_depth(void)
{
  return g_depth;
}

/* test hook: the stream is busy for `usec` (a kernel that takes its time) */
void fake_hip_stream_busy(void* stream, unsigned usec)
{
  S(static_cast<hipStream_t>(stream))->push([usec] { std::this_thread::sleep_for(std::chrono::microseconds(usec)); });
}

/* ---- HIP ---- */
hipError_t hipSetDevice(int d)
{
  return d >= 0 && d < kMaxRanks ? hipSuccess : hipErrorInvalidDevice;
}
const char* hipGetErrorString(hipError_t e)
{
  return e == hipSuccess ? "no error" : "fake HIP error";
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
  *s = reinterpret_cast<hipStream_t>(new FakeStream);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
  delete reinterpret_cast<FakeStream*>(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
  S(s)->sync();
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned)
{
  *e = reinterpret_cast<hipEvent_t>(new FakeEvent);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e)
{
  return hipEventCreateWithFlags(e, 0);
}
hipError_t hipEventDestroy(hipEvent_t e)
{
  delete E(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e_, hipStream_t s)
{
  FakeEvent* e = E(e_);
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> l(e->m);
    ticket = ++e->recorded;
  }
  S(s)->push([e, ticket] {
    {
      std::lock_guard<std::mutex> l(e->m);
      e->done = ticket;
      e->when = std::chrono::steady_clock::now();
    }
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e_, unsigned)
{
  FakeEvent* e = E(e_);
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> l(e->m);
    ticket = e->recorded; // the latest record at the time of the call; never recorded: no wait
  }
  if (ticket)
    S(s)->push([e, ticket] {
      std::unique_lock<std::mutex> l(e->m);
      e->cv.wait(l, [&] { return e->done >= ticket; });
    });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
  *ms = std::chrono::duration<float, std::milli>(E(b)->when - E(a)->when).count();
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n)
{
  *p = malloc(n);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p)
{
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s)
{
  S(s)->push([dst, src, n] { memcpy(dst, src, n); });
  return hipSuccess;
}

/* ---- RCCL ---- */
const char* ncclGetErrorString(ncclResult_t r)
{
  return r == ncclSuccess ? "no error" : "fake RCCL error";
}
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%ld", int(getpid()),
           long(std::chrono::steady_clock::now().time_since_epoch().count() % 1000000007L));
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
    return ncclInvalidArgument;
  { // FAKE_RCCL_STALL_INIT=<rank>: that rank's bootstrap never returns on the first attempt (tests of the restart)
    const char* stall = getenv("FAKE_RCCL_STALL_INIT");
    const char* attempt = getenv("FAKE_RCCL_ATTEMPT");
    if (stall && atoi(stall) == rank && !(attempt && attempt[0] == '1'))
      for (;;)
        pause();
  }
  ncclComm* c = new ncclComm;
  c->rank = rank;
  c->world = world;
  snprintf(c->name, sizeof c->name, "%.60s", id.internal);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0)
    return ncclSystemError;
  c->sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->sh == MAP_FAILED)
    return ncclSystemError;
  c->sh->alive.fetch_add(1);
  c->sh->joined.fetch_add(1); // (a fresh segment reads zero)
  for (int i = 0; c->sh->joined.load() < world; i++)
  { // collective: every rank has to arrive
    if (i > 200000)
      return ncclSystemError;
    usleep(100);
  }
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
  if (g_depth)
    die("ncclCommDestroy with a group left open");
  if (c->sh->alive.fetch_sub(1) == 1)
    shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}
static int g_aborts = 0;
extern "C" int fake_rccl_aborts(void)
{
  return g_aborts;
}
ncclResult_t ncclCommAbort(ncclComm_t c)
{ // what a caller does with a communicator whose last collective is incomplete: no check of open groups
  g_aborts++;
  if (c->sh->alive.fetch_sub(1) == 1)
    shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n)
{
  *n = c->sh->joined.load(); // who really joined, not what the caller asked for
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r)
{
  *r = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int* d)
{
  *d = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart()
{
  g_depth++;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
  if (g_depth <= 0)
    die("ncclGroupEnd without ncclGroupStart");
  if (--g_depth == 0 && !g_ops.empty())
  {
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclComm* c = g_comm;
    S(g_stream)->push([c, ops] { run_group(c, ops); });
  }
  if (g_depth == 0)
  {
    g_ops.clear();
    g_comm = nullptr;
  }
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
  return add_op(true, buf, count, t, peer, c, s);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
  const int k = ++g_recv_calls;
  if (fail_recv_at() && k == fail_recv_at())
    return ncclInternalError;
  return add_op(false, buf, count, t, peer, c, s);
}
ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t s)
{
  if (count != 1 || t != ncclDouble || op != ncclMax)
    die("the double only knows the barrier's all-reduce (one double, max)");
  const uint64_t gen = c->ar_gen++;
  S(s)->push([c, in, out, gen] {
    const int p = int(gen & 1);
    c->sh->vals[p][c->rank] = *static_cast<const double*>(in);
    c->sh->arrive[p].fetch_add(1);
    while (c->sh->arrive[p].load() < uint64_t(c->world) * (gen / 2 + 1))
      std::this_thread::yield();
    double m = c->sh->vals[p][0];
    for (int r = 1; r < c->world; r++)
      m = std::max(m, c->sh->vals[p][r]);
    *static_cast<double*>(out) = m;
  });
  return ncclSuccess;
}
}
