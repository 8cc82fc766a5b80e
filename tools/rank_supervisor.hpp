/*
 * rank_supervisor.hpp -- the parent of a whole-node run: forks one rank per GPU BEFORE anything touches HIP, and
 * starts all of them over ONCE, as fresh children, when not every rank has reported "communicator up" in time.
 *
 * Why: RCCL's bootstrap (ncclCommInitRank) stalls about once in 20 launches on this pool -- no error, no output,
 * every rank waits inside the collective until its own watchdog ends it.  A second try from fresh processes has
 * always come up; the first 8-GPU run must not be the one launch in 20 that ends as a watchdog exit.  The parent
 * never initialises the GPU (a process that has must not fork workers on this pool), so killing and re-forking the
 * ranks is safe.
 *
 * Used by tools/node_bench.cpp; tests/cpp/gather_double/world_n.cpp runs it against the HIP + RCCL test double with
 * an ncclCommInitRank that stalls on the first attempt (tests/test_gather_double.py).  bench.py has the same logic
 * per rank in Python (_supervise_rank).  Measurement / launch infrastructure, not part of the decoder library.
 */
#pragma once

#include <poll.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>

namespace fmd_launch
{

/* rank_main(rank, attempt, up_fd): runs in the child; writes one byte to up_fd once its communicator is up (every rank
 * does: the communicator's creation is collective) and returns the rank's exit code.
 * Returns 0 when every rank of the final attempt exited with 0.  *attempts_out = 1 or 2. */
inline int run_ranks(int n, int up_timeout_s, unsigned watchdog_s,
                     const std::function<int(int rank, int attempt, int up_fd)>& rank_main, int* attempts_out = nullptr)
{
  using clk = std::chrono::steady_clock;
  for (int attempt = 0; attempt < 2; attempt++)
  {
    int fds[2];
    if (pipe(fds) != 0)
      return 1;
    std::vector<pid_t> kids;
    for (int r = 0; r < n; r++)
    {
      const pid_t p = fork();
      if (p == 0)
      {
        close(fds[0]);
        alarm(watchdog_s); // no rank outlives its watchdog, whatever happens to its parent
        _exit(rank_main(r, attempt, fds[1]));
      }
      kids.push_back(p);
    }
    close(fds[1]);
    if (attempts_out)
      *attempts_out = attempt + 1;
    // phase 1: until every rank has said "up", a rank has exited, or the time is over
    int up = 0, exited = 0, bad = 0;
    std::vector<bool> done(kids.size(), false);
    auto reap = [&](bool block) {
      for (size_t k = 0; k < kids.size(); k++)
        if (!done[k])
        {
          int st = 0;
          const pid_t p = waitpid(kids[k], &st, block ? 0 : WNOHANG);
          if (p == kids[k])
          {
            done[k] = true;
            exited++;
            bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
          }
        }
    };
    const auto t_end = clk::now() + std::chrono::seconds(up_timeout_s);
    bool pipe_open = true;
    while (up < n && exited == 0 && clk::now() < t_end)
    {
      struct pollfd pf = {fds[0], POLLIN, 0};
      if (pipe_open && poll(&pf, 1, 100) > 0)
      {
        char buf[64];
        const ssize_t got = read(fds[0], buf, sizeof buf);
        if (got > 0)
          up += int(got);
        else if (got == 0)
          pipe_open = false; // every writer has gone
      }
      else if (!pipe_open)
        usleep(100 * 1000);
      reap(false);
    }
    close(fds[0]);
    if (up < n && attempt == 0 && (exited == 0 || bad > 0))
    { // stalled (or a rank failed before the communicator was up): end exactly the children started here, start over
      fprintf(stderr, "rank supervisor: %d of %d ranks reported their communicator up within %d s -- "
                      "ending the ranks and starting all of them again, once\n", up, n, up_timeout_s);
      for (size_t k = 0; k < kids.size(); k++)
        if (!done[k])
          kill(kids[k], SIGKILL);
      reap(true);
      continue;
    }
    // phase 2: the run itself.  A rank that fails leaves its peers waiting for it: end them instead of hanging.
    while (exited < n)
    {
      const int before = exited;
      reap(false);
      if (bad > 0)
      {
        for (size_t k = 0; k < kids.size(); k++)
          if (!done[k])
            kill(kids[k], SIGKILL);
        reap(true);
      }
      else if (exited == before)
        usleep(20 * 1000);
    }
    return bad ? 1 : 0;
  }
  return 1;
}

} // namespace fmd_launch
