/*
 * world_n.cpp -- fmd_gather_step with a world of N ranks in the build container: the product's
 * csrc/fmd_gather.hip compiled with g++ and linked against the test double of HIP + RCCL
 * (fake_hip_rccl.cpp) instead of the real libraries.  N processes (forked before anything else), rotating
 * buffers like tools/node_bench.cpp, 40 steps (the 16-event ring wraps twice); rank 0 checks every byte
 * it received against what rank r must have sent for that step -- so the send / receive pairing, the offsets
 * d_all_audio + r * audio_floats and d_all_rds + r * rds_rows * 4, the ordering behind the caller's stream
 * and the lagged waits have all run at least once with a real peer.
 *
 *   world_n <ranks> <steps>          exit code 0 = every rank succeeded
 *   FAKE_RCCL_FAIL_RECV=1 world_n 2 1    rank 0's first receive fails: the step must report it and leave no
 *                                        group open
 *   WORLD_N_ROTATE=1 world_n 3 40        step i is gathered to rank i % world (fmd_gather_step_root): every rank is
 *                                        a root in turn, has its own part produced in place in its slot of its
 *                                        receive buffers, and checks every byte of the steps it received
 *   FAKE_RCCL_STALL_INIT=<rank> WORLD_N_UP_TIMEOUT=2 world_n 3 8
 *                                        that rank's first ncclCommInitRank never returns (the pool's stalled
 *                                        bootstrap): the parent (tools/rank_supervisor.hpp, what tools/node_bench
 *                                        uses) ends all ranks and starts them again, once; the run then completes
 */
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/fmd_gather.h"
#include "../../../tools/rank_supervisor.hpp"

extern "C" int fake_rccl_group_depth(void);
extern "C" int fake_rccl_aborts(void);
extern "C" int hipStreamCreateWithFlags(void**, unsigned);
extern "C" int hipStreamSynchronize(void*);
extern "C" int hipMemcpyAsync(void*, const void*, size_t, int, void*);
extern "C" void fake_hip_stream_busy(void*, unsigned);

/* what the product library would bring (libfmd_hip.so); the test passes batch = NULL, so never called */
extern "C" int fmd_batch_export_rds_device(fmd_batch*, int32_t*, unsigned, unsigned, int, void*)
{
  abort();
}
extern "C" const char* fmd_last_error(void)
{
  return "";
}

#define CHECK(x)                                                                                       \
  do                                                                                                   \
  {                                                                                                    \
    if (!(x))                                                                                          \
    {                                                                                                  \
      fprintf(stderr, "rank %d: %s failed (%s)\n", rank, #x, fmd_gather_last_error());                \
      _exit(1);                                                                                        \
    }                                                                                                  \
  } while (0)

static float audio_value(int r, int step, size_t j)
{
  return float(r * 1000000 + step * 1000 + int(j % 997));
}
static int32_t rds_value(int r, int step, size_t k)
{
  return int32_t((r << 24) | (step << 12) | int(k & 0xFFF));
}

static int rank_main(int rank, int world, int steps, const uint8_t* id, bool expect_failure, int attempt, int up_fd)
{
  setenv("FAKE_RCCL_ATTEMPT", attempt ? "1" : "0", 1); // (the double's stalling ncclCommInitRank reads it)
  const size_t AFL = 50021; // floats per rank and step: larger than the double's rings, not a round number
  const unsigned ROWS = 37;
  const int NBUF = 6;
  fmd_gather* g = nullptr;
  CHECK(fmd_gather_create(id, rank, world, rank, AFL, ROWS, &g) == FMD_OK);
  {
    const char u = 'U';
    (void)!write(up_fd, &u, 1);
    close(up_fd);
  }
  fmd_gather_info_t inf;
  CHECK(fmd_gather_info(g, &inf) == FMD_OK);
  CHECK(inf.ranks_seen == world && inf.rank == rank && inf.world_asked == world && inf.steps_issued == 0);
  void* st = nullptr;
  CHECK(hipStreamCreateWithFlags(&st, 0) == 0);
  const bool rotate = getenv("WORLD_N_ROTATE") != nullptr;
  auto root_of = [&](int step) { return rotate ? step % world : 0; };
  const bool receives = rotate || rank == 0;
  std::vector<float> audio(size_t(NBUF) * AFL), all_a(receives ? size_t(NBUF) * world * AFL : 0);
  std::vector<int32_t> rds(size_t(NBUF) * ROWS * 4), all_r(receives ? size_t(NBUF) * world * ROWS * 4 : 0);
  std::vector<float> stage_a(AFL);
  std::vector<int32_t> stage_r(size_t(ROWS) * 4);
  const bool inplace = getenv("WORLD_N_INPLACE") != nullptr || rotate;
  long checked = 0;
  auto verify = [&](int step) { // the step's root: what every rank sent in `step`
    if (root_of(step) != rank)
      return;
    const int s = step % NBUF;
    for (int r = 0; r < world; r++)
    {
      const float* a = &all_a[(size_t(s) * world + r) * AFL];
      for (size_t j = 0; j < AFL; j++)
        if (a[j] != audio_value(r, step, j))
        {
          fprintf(stderr, "rank 0: audio of rank %d, step %d, float %zu: %g != %g\n", r, step, j, a[j], audio_value(r, step, j));
          _exit(1);
        }
      const int32_t* q = &all_r[(size_t(s) * world + r) * ROWS * 4];
      for (size_t k = 0; k < size_t(ROWS) * 4; k++)
        if (q[k] != rds_value(r, step, k))
        {
          fprintf(stderr, "rank 0: records of rank %d, step %d, word %zu\n", r, step, k);
          _exit(1);
        }
      checked++;
    }
  };
  for (int i = 0; i < steps; i++)
  {
    const int s = i % NBUF;
    // the step that last used this slot (NBUF steps ago) must have read it / filled it: all but the NBUF - 1 youngest
    CHECK(fmd_gather_wait_lagged(g, NBUF - 1, st) == FMD_OK);
    CHECK(hipStreamSynchronize(st) == 0);
    if (i >= NBUF)
      verify(i - NBUF);
    // the step's outputs are produced ON THE CALLER'S STREAM, late (a busy stream, then a copy from staging
    // buffers): a gather that does not order its own stream behind the caller's sends the poison below
    for (size_t j = 0; j < AFL; j++)
      stage_a[j] = audio_value(rank, i, j);
    for (size_t k = 0; k < size_t(ROWS) * 4; k++)
      stage_r[k] = rds_value(rank, i, k);
    // (WORLD_N_INPLACE: rank 0 has its outputs produced in its part of the receive buffers, like tools/node_bench)
    const int root = root_of(i);
    float* const my_a = (rank == root && inplace) ? &all_a[(size_t(s) * world + root) * AFL] : &audio[size_t(s) * AFL];
    int32_t* const my_r = (rank == root && inplace) ? &all_r[(size_t(s) * world + root) * ROWS * 4] : &rds[size_t(s) * ROWS * 4];
    if (rank == root)
    { // poison the receive slot: stale data from NBUF steps ago must not pass for this step's
      memset(&all_a[size_t(s) * world * AFL], 0xFF, size_t(world) * AFL * 4);
      memset(&all_r[size_t(s) * world * ROWS * 4], 0xFF, size_t(world) * ROWS * 16);
    }
    memset(my_a, 0xEE, AFL * 4);
    memset(my_r, 0xEE, size_t(ROWS) * 16);
    fake_hip_stream_busy(st, 1500);
    CHECK(hipMemcpyAsync(my_a, stage_a.data(), AFL * 4, 0, st) == 0);
    CHECK(hipMemcpyAsync(my_r, stage_r.data(), size_t(ROWS) * 16, 0, st) == 0);
    const int rc = fmd_gather_step_root(g, root, nullptr, 0, 0, my_a, my_r,
                                        rank == root ? &all_a[size_t(s) * world * AFL] : nullptr,
                                        rank == root ? &all_r[size_t(s) * world * ROWS * 4] : nullptr, st);
    if (expect_failure && rank == 0)
    {
      CHECK(rc < 0 && strstr(fmd_gather_last_error(), "ncclRecv") != nullptr);
      CHECK(fake_rccl_group_depth() == 0); // the failed step closed its group
      // ... and the gather takes no further step (its peers are inside a half-done one); destroy aborts
      const int rc2 = fmd_gather_step(g, nullptr, 0, 0, &audio[0], &rds[0], &all_a[0], &all_r[0], st);
      CHECK(rc2 == FMD_ERR_STATE && strstr(fmd_gather_last_error(), "tear the communicator down") != nullptr);
      fmd_gather_destroy(g);
      CHECK(fake_rccl_aborts() == 1);
      _exit(0);
    }
    CHECK(rc >= 0);
  }
  if (expect_failure)
    _exit(0); // (a sender whose peer gave up: nothing more to check)
  CHECK(fmd_gather_wait(g, st) == FMD_OK && hipStreamSynchronize(st) == 0);
  for (int i = steps > NBUF ? steps - NBUF : 0; i < steps; i++)
    verify(i);
  double mx = -1.0;
  CHECK(fmd_gather_barrier(g, 10.0 + rank, &mx) == FMD_OK && mx == 10.0 + (world - 1));
  CHECK(fmd_gather_barrier(g, 5.0 - rank, &mx) == FMD_OK && mx == 5.0);
  CHECK(fmd_gather_ms_per_step(g) >= 0.0f && fmd_gather_ms_per_step(g) < 0.0f); // the second query has nothing to report
  CHECK(fmd_gather_info(g, &inf) == FMD_OK && inf.steps_issued == uint64_t(steps) && inf.ranks_seen == world);
  if (rotate) // every rank was a root in turn
    printf("{\"rank\": %d, \"world\": %d, \"rank_step_messages_checked\": %ld, \"rccl_ranks_seen\": %d}\n", rank, world,
           checked, inf.ranks_seen);
  else if (rank == 0)
    printf("{\"world\": %d, \"steps\": %d, \"rank_step_messages_checked\": %ld, \"rccl_ranks_seen\": %d}\n", world, steps,
           checked, inf.ranks_seen);
  fflush(stdout);
  fmd_gather_destroy(g);
  return 0;
}

int main(int argc, char** argv)
{
  const int world = argc > 1 ? atoi(argv[1]) : 2, steps = argc > 2 ? atoi(argv[2]) : 40;
  const bool expect_failure = getenv("FAKE_RCCL_FAIL_RECV") != nullptr;
  uint8_t id[2][FMD_GATHER_ID_BYTES]; // (the double's id is a name: made before the ranks exist; one per attempt)
  if (fmd_gather_unique_id(id[0]) != FMD_OK || (usleep(1000), fmd_gather_unique_id(id[1])) != FMD_OK)
    return 2;
  const int up_timeout = getenv("WORLD_N_UP_TIMEOUT") ? atoi(getenv("WORLD_N_UP_TIMEOUT")) : 60;
  int attempts = 0;
  const int bad = fmd_launch::run_ranks(
      world, up_timeout, 120,
      [&](int r, int attempt, int up_fd) { return rank_main(r, world, steps, id[attempt], expect_failure, attempt, up_fd); },
      &attempts);
  fprintf(stderr, "world_n: attempts %d\n", attempts);
  shm_unlink(reinterpret_cast<const char*>(id[1]));
  shm_unlink(reinterpret_cast<const char*>(id[0])); // (ranks that left through _exit did not remove the double's segment)
  return bad ? 1 : 0;
}
