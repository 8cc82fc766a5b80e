/*
 * node_bench.cpp -- the whole-node timed loop without Python: one process per GPU, every rank decodes
 * its own 8192 synthetic stations (include/fmd.h) and rank 0 gathers audio and RDS records of every step
 * over RCCL (include/fmd_gather.h).  The same loop as bench.py's (calls overlapped, outputs consumed
 * LAG steps late), same JSON keys; bench.py stays the driver's entry point.
 *
 *   tools/node_bench --gpus N [--steps K] [--warmup W] [--channels C] [--watchdog seconds] [--up-timeout seconds]
 *                    [--gather-root 0|rotate] [--own-stream 0|1|2|3] [--verify]
 *
 * The parent forks the N ranks BEFORE anything touches HIP; rank 0 writes the communicator's id into a
 * file the others wait for.  Build: make -C pvr.rtl.radiofm_amd/csrc ../../tools/node_bench
 */
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/fmd_gather.h"
#include "rank_supervisor.hpp"
#include "fmsig.h"
#include "fmsig_core.h"

extern "C" int fmsig_device_generate(const void*, const void*, unsigned, unsigned, uint64_t, unsigned, void*, size_t, void*);
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "rank %d: %s failed (%s / %s)\n", rank, #x, fmd_last_error(), fmd_gather_last_error()); _exit(1); } } while (0)

/* station g of the synthetic node-wide workload (tools/fmsig_py.channel_params) */
static void make_station(double fs, unsigned g, fmsig_chan* ch, uint8_t* dbits)
{
  fmsig_params p;
  fmsig_default(&p, fs);
  p.noise_sigma = 0.01;
  p.seed = 1000 + g;
  p.f_left = 400.0 + 13.0 * (g % 97);
  p.f_right = 2500.0 + 7.0 * (g % 211);
  p.pi = uint16_t(0x1000 + (g % 0xE000));
  char ps[16];
  snprintf(ps, sizeof ps, "C%07u", g % 10000000u);
  memcpy(p.ps, ps, 8);
  *ch = fmsig_chan{1.0 / p.fs, p.f_offset, p.dev, p.amp, p.a_mono, p.a_stereo, p.a_pilot, p.a_rds, p.f_left, p.f_right, p.noise_sigma, p.seed};
  fmsig_rds_dbits(&p, dbits);
}

static int rank_main(int rank, int world, int K, int W, unsigned C, const std::string& idfile, bool verify, bool rotate,
                     int own_stream, int up_fd)
{
  // --gather-root rotate: step i is gathered to rank i % world (fmd_gather_step_root) instead of rank 0
  auto root_of = [&](int step) { return rotate ? step % world : 0; };
  const bool receives = rotate || rank == 0;
  const double fs = 2.4e6;
  const unsigned N = 65536, D = 11, RING = 10, LAG = 3, NBUF = LAG + 3;
  CHECK(hipSetDevice(rank) == hipSuccess);
  uint8_t id[FMD_GATHER_ID_BYTES];
  if (rank == 0)
  {
    CHECK(fmd_gather_unique_id(id) == FMD_OK);
    FILE* f = fopen((idfile + ".tmp").c_str(), "wb");
    CHECK(f && fwrite(id, 1, sizeof id, f) == sizeof id);
    fclose(f);
    rename((idfile + ".tmp").c_str(), idfile.c_str());
  }
  else
  {
    FILE* f = nullptr;
    for (int i = 0; i < 600 && !(f = fopen(idfile.c_str(), "rb")); i++)
      usleep(100000);
    CHECK(f && fread(id, 1, sizeof id, f) == sizeof id);
    fclose(f);
  }
  // stations rank * C .. rank * C + C - 1 (tools/fmsig_py.channel_params), generated on the device
  std::vector<fmsig_chan> ch(C);
  std::vector<uint8_t> dbits(size_t(C) * FMSIG_RDS_PERIOD_BITS);
  for (unsigned c = 0; c < C; c++)
    make_station(fs, unsigned(rank) * C + c, &ch[c], &dbits[size_t(c) * FMSIG_RDS_PERIOD_BITS]);
  void *d_ch, *d_bits;
  float* iq;
  CHECK(hipMalloc(&d_ch, ch.size() * sizeof(fmsig_chan)) == hipSuccess && hipMalloc(&d_bits, dbits.size()) == hipSuccess);
  CHECK(hipMemcpy(d_ch, ch.data(), ch.size() * sizeof(fmsig_chan), hipMemcpyHostToDevice) == hipSuccess);
  CHECK(hipMemcpy(d_bits, dbits.data(), dbits.size(), hipMemcpyHostToDevice) == hipSuccess);
  CHECK(hipMalloc(reinterpret_cast<void**>(&iq), size_t(RING) * C * N * 8) == hipSuccess);
  for (unsigned r = 0; r < RING; r++)
    CHECK(fmsig_device_generate(d_ch, d_bits, FMSIG_RDS_PERIOD_BITS, C, uint64_t(r) * N, N, iq + size_t(r) * C * N * 2, N, nullptr) == 0);
  CHECK(hipDeviceSynchronize() == hipSuccess);

  // (input first, then the decoder: the allocation order bench.py has -- with the decoder's buffers in front of the
  // 43 GB input ring the same loop measured 3-4 % slower, round 6)
  fmd_params par{fs, -0.15 * fs, 48000.0, 15000.0, D, 0, 0, 0, FMD_FIR_SEQUENTIAL};
  fmd_batch* b = nullptr;
  CHECK(fmd_batch_create(&par, C, nullptr, rank, nullptr, nullptr, &b) == FMD_OK);
  CHECK(fmd_batch_set_concurrency(b, 2) == FMD_OK);
  const size_t stride = (fmd_batch_max_audio_floats(b, N) + 63) / 64 * 64, afl = stride * C;
  fmd_gather* g = nullptr;
  CHECK(fmd_gather_create(id, rank, world, rank, afl, C, &g) == FMD_OK);
  if (up_fd >= 0)
  { // the communicator is up: tell the parent (tools/rank_supervisor.hpp starts all ranks over when one never does)
    const char u = 'U';
    (void)!write(up_fd, &u, 1);
    close(up_fd);
  }
  float *audio = nullptr, *all_a = nullptr;
  int32_t *rds = nullptr, *all_r = nullptr;
  if (rotate || rank != 0) // (a root's outputs are produced in place, in its part of its receive buffers: audio_of / rds_of)
    CHECK(hipMalloc(reinterpret_cast<void**>(&audio), NBUF * afl * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&rds), size_t(NBUF) * C * 16) == hipSuccess);
  if (receives)
    CHECK(hipMalloc(reinterpret_cast<void**>(&all_a), size_t(NBUF) * world * afl * 4) == hipSuccess &&
          hipMalloc(reinterpret_cast<void**>(&all_r), size_t(NBUF) * world * C * 16) == hipSuccess);
  hipStream_t st = nullptr;
  // The caller's stream is the null stream, like bench.py's: the same loop on a created stream of the default priority
  // runs 3-4 % slower (283 200-285 500 against 289 300-293 000 MS/s, three alternating runs on one box; bench.py
  // --side-stream: the same) although the batch's probe finds no hardware queue shared with it
  // (fmd_batch_debug_stream_conflicts: 0); a created stream at the least or the greatest priority does not (290 500-
  // 293 000).  "--own-stream 1 | 2 | 3": default / least / greatest priority (docs/MEASUREMENTS.md, round 6).
  if (own_stream == 1)
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
  else if (own_stream >= 2)
  { // 2: at the least priority (what the decoder's heavy and light streams have), 3: at the greatest
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess);
    CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, own_stream == 2 ? lo : hi) == hipSuccess);
  }
  int submitted = -1, finalized = -1;
  unsigned nf = 0;
  std::vector<unsigned> nfs;
  // a step's root has its own outputs produced in place: slot s of its audio / record buffers is its part (the
  // rank-th) of slot s of its receive buffers, and the gather has nothing to copy for it
  auto audio_of = [&](int i) {
    const int s = i % int(NBUF);
    return rank == root_of(i) ? all_a + (size_t(s) * world + rank) * afl : audio + size_t(s) * afl;
  };
  auto rds_of = [&](int i) {
    const int s = i % int(NBUF);
    return rank == root_of(i) ? all_r + (size_t(s) * world + rank) * C * 4 : rds + size_t(s) * C * 4;
  };
  auto finalize = [&](int lag) { // outputs of step finalized + 1: on their way to the step's root
    const int i = ++finalized, s = i % int(NBUF);
    CHECK(fmd_gather_step_root(g, root_of(i), b, lag, unsigned(rank) * C, audio_of(i), rds_of(i),
                               all_a ? all_a + size_t(s) * world * afl : nullptr,
                               all_r ? all_r + size_t(s) * world * C * 4 : nullptr, st) >= 0);
  };
  auto step = [&](int i) {
    CHECK(fmd_gather_wait_lagged(g, NBUF - LAG - 1, st) == FMD_OK); // the gather that last read this slot's buffers
    CHECK(fmd_batch_process_device(b, iq + size_t(i % int(RING)) * C * N * 2, N, N, audio_of(i), stride, &nf, st) == FMD_OK);
    submitted = i;
    nfs.push_back(nf); // (audio floats per channel of step i: 2620 / 2622 at 2.4 MS/s)
    if (i - int(LAG) > finalized)
    {
      CHECK(fmd_batch_wait_lagged(b, LAG, st) >= 0);
      while (finalized < i - int(LAG))
        finalize(LAG);
      // the host blocks on the call LAG steps back (the export of its groups is the last thing on the caller's
      // stream; the gather's own stream is not waited for): without it nothing stops the host before the decoder's
      // limit of eight calls in flight, and the device runs 2.5 % slower behind a queue that deep (11 % over 20 steps)
      CHECK(hipStreamSynchronize(st) == hipSuccess);
    }
  };
  auto drain = [&]() {
    while (finalized < submitted)
    {
      const int lag = submitted - (finalized + 1);
      CHECK(fmd_batch_wait_lagged(b, lag, st) >= 0);
      finalize(lag);
    }
    CHECK(fmd_batch_wait(b, st) >= 0 && fmd_gather_wait(g, st) == FMD_OK && hipStreamSynchronize(st) == hipSuccess);
  };
  for (int i = 0; i < W; i++)
    step(i);
  drain();
  // what the communicator says about itself: N processes that each ran a world of one would read 1 here
  fmd_gather_info_t inf;
  CHECK(fmd_gather_info(g, &inf) == FMD_OK);
  double seen_short = 0.0;
  CHECK(fmd_gather_barrier(g, double(world - inf.ranks_seen), &seen_short) == FMD_OK);
  CHECK(seen_short == 0.0 && inf.rank == rank);
  /* --verify: what a root (rank 0; every rank where the root rotates) received from every rank in the last warm-up
   * steps against a recomputation of four stations per rank in a batch of its own (same generator, same call
   * sequence), bit for bit: audio rows and RDS records.  Every rank learns the verdict (the barriers' maxima) and
   * leaves with code 4 on a mismatch. */
  std::vector<int> rank_ok(size_t(world), 1);
  int verified_steps = 0;
  if (verify)
  {
    double bad = 0.0;
    if (receives)
    {
      std::vector<unsigned> picks = {0u, 1u, C / 2, C - 1};
      std::sort(picks.begin(), picks.end());
      picks.erase(std::unique(picks.begin(), picks.end()), picks.end());
      const unsigned np = unsigned(picks.size()), nv = unsigned(world) * np;
      std::vector<fmsig_chan> vch(nv);
      std::vector<uint8_t> vbits(size_t(nv) * FMSIG_RDS_PERIOD_BITS);
      for (int r = 0; r < world; r++)
        for (unsigned k = 0; k < np; k++)
          make_station(fs, unsigned(r) * C + picks[k], &vch[size_t(r) * np + k], &vbits[(size_t(r) * np + k) * FMSIG_RDS_PERIOD_BITS]);
      void *d_vch, *d_vbits;
      float *viq, *vaudio;
      int32_t* vrds;
      const unsigned vrows = 4 * nv;
      CHECK(hipMalloc(&d_vch, vch.size() * sizeof(fmsig_chan)) == hipSuccess && hipMalloc(&d_vbits, vbits.size()) == hipSuccess);
      CHECK(hipMemcpy(d_vch, vch.data(), vch.size() * sizeof(fmsig_chan), hipMemcpyHostToDevice) == hipSuccess);
      CHECK(hipMemcpy(d_vbits, vbits.data(), vbits.size(), hipMemcpyHostToDevice) == hipSuccess);
      CHECK(hipMalloc(reinterpret_cast<void**>(&viq), size_t(nv) * N * 8) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&vaudio), size_t(nv) * stride * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&vrds), size_t(vrows) * 16) == hipSuccess);
      fmd_batch* vb = nullptr;
      CHECK(fmd_batch_create(&par, nv, nullptr, rank, nullptr, nullptr, &vb) == FMD_OK);
      std::vector<float> ha(stride), hb(stride);
      std::vector<int32_t> hv(size_t(vrows) * 4), hg(size_t(C) * 4);
      const int V = std::min(W, int(NBUF));
      for (int i = 0; i < W; i++)
      {
        unsigned vnf = 0;
        CHECK(fmsig_device_generate(d_vch, d_vbits, FMSIG_RDS_PERIOD_BITS, nv, uint64_t(i % int(RING)) * N, N, viq, N, st) == 0);
        CHECK(fmd_batch_process_device(vb, viq, N, N, vaudio, stride, &vnf, st) == FMD_OK);
        CHECK(fmd_batch_wait(vb, st) >= 0);
        CHECK(fmd_batch_export_rds_device(vb, vrds, vrows, 0, 0, st) >= 0);
        CHECK(hipStreamSynchronize(st) == hipSuccess);
        if (i < W - V || root_of(i) != rank)
          continue;
        verified_steps++;
        const int s = i % int(NBUF);
        CHECK(hipMemcpy(hv.data(), vrds, hv.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
        for (int r = 0; r < world; r++)
        {
          CHECK(hipMemcpy(hg.data(), all_r + (size_t(s) * world + r) * C * 4, hg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
          for (unsigned k = 0; k < np; k++)
          {
            CHECK(hipMemcpy(ha.data(), all_a + (size_t(s) * world + r) * afl + size_t(picks[k]) * stride, stride * 4, hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(hipMemcpy(hb.data(), vaudio + (size_t(r) * np + k) * stride, stride * 4, hipMemcpyDeviceToHost) == hipSuccess);
            if (vnf != nfs[size_t(i)] || memcmp(ha.data(), hb.data(), size_t(vnf) * 4) != 0)
              rank_ok[size_t(r)] = 0;
            // records: [channel + 1, call index, blocks]; the same groups of the same call on both sides
            std::vector<std::array<int32_t, 3>> got, want;
            for (unsigned q = 0; q < C; q++)
              if (hg[size_t(q) * 4] == int32_t(unsigned(r) * C + picks[k] + 1))
                got.push_back({hg[size_t(q) * 4 + 1], hg[size_t(q) * 4 + 2], hg[size_t(q) * 4 + 3]});
            for (unsigned q = 0; q < vrows; q++)
              if (hv[size_t(q) * 4] == int32_t(unsigned(r) * np + k + 1))
                want.push_back({hv[size_t(q) * 4 + 1], hv[size_t(q) * 4 + 2], hv[size_t(q) * 4 + 3]});
            std::sort(got.begin(), got.end());
            std::sort(want.begin(), want.end());
            if (got != want)
              rank_ok[size_t(r)] = 0;
          }
        }
      }
      fmd_batch_destroy(vb);
      (void)hipFree(d_vch); (void)hipFree(d_vbits); (void)hipFree(viq); (void)hipFree(vaudio); (void)hipFree(vrds);
      for (int r = 0; r < world; r++)
        if (!rank_ok[size_t(r)])
        {
          fprintf(stderr, "node_bench --verify: what rank %d received from rank %d differs from the recomputation\n", rank, r);
          bad = 1.0;
        }
    }
    double worst = 0.0, steps_all = 0.0;
    CHECK(fmd_gather_barrier(g, bad, &worst) == FMD_OK);
    for (int r = 0; r < world; r++)
    { // every rank's view of every sender (a rotating root: the roots' verdicts combined)
      double w = 0.0;
      CHECK(fmd_gather_barrier(g, rank_ok[size_t(r)] ? 0.0 : 1.0, &w) == FMD_OK);
      rank_ok[size_t(r)] = w == 0.0;
    }
    CHECK(fmd_gather_barrier(g, double(verified_steps), &steps_all) == FMD_OK);
    if (rotate)
      verified_steps = int(steps_all); // (the most steps any one root verified)
    if (worst != 0.0)
      _exit(4);
  }
  (void)fmd_gather_ms_per_step(g);
  CHECK(fmd_gather_barrier(g, 0.0, nullptr) == FMD_OK);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = W; i < W + K; i++)
    step(i);
  drain();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), dmax = dt;
  CHECK(fmd_gather_barrier(g, dt, &dmax) == FMD_OK); // the slowest rank's time
  const float gms = fmd_gather_ms_per_step(g);
  std::string vjson = "null";
  if (verify)
  {
    vjson = "{\"steps\": " + std::to_string(verified_steps) + ", \"ranks\": " + std::to_string(world) + ", \"per_rank_ok\": [";
    for (int r = 0; r < world; r++)
      vjson += std::string(r ? ", " : "") + (rank_ok[size_t(r)] ? "true" : "false");
    bool all_ok = true;
    for (int r = 0; r < world; r++)
      all_ok = all_ok && rank_ok[size_t(r)] != 0;
    vjson += std::string("], \"ok\": ") + (all_ok ? "true" : "false") + "}";
  }
  if (rank == 0)
    printf("{\"metric\": \"IQ MS/s demodulated (whole node) + achieved HBM GB/s on FIR stage\", \"value\": %.1f, \"unit\": \"MS/s\", "
           "\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", "
           "\"vs_baseline\": null, \"dtype\": \"f32\", \"data\": \"synthetic\", \"config\": {\"workload\": \"BASELINE configs[3] per-GPU shard: "
           "%u independent FM stereo+RDS channels/GPU @2.4 MS/s, 65536 IQ/channel/step (tools/node_bench.cpp: C++ host, RCCL gather to %s)\", "
           "\"audio_floats_per_channel_step\": %u, \"gather_ms_per_step_rank0\": %.4f}, \"rccl_ranks_seen\": %d, \"verify\": %s}\n",
           double(world) * C * N * K / dmax / 1e6, world, K, W, dmax / K * 1e3, C, rotate ? "a rotating root: step i to rank i % world" : "rank 0",
           nf, gms, inf.ranks_seen, vjson.c_str());
  fflush(stdout); // (the rank leaves through _exit)
  fmd_gather_destroy(g);
  fmd_batch_destroy(b);
  return 0;
}

int main(int argc, char** argv)
{
  int gpus = 1, K = 40, W = 8;
  unsigned C = 8192;
  bool verify = false, rotate = false;
  int own_stream = 0;
  unsigned watchdog = 900; // seconds after which a rank ends itself (a stalled RCCL bootstrap must not hang the node)
  int up_timeout = 90;     // seconds for every rank's communicator to be up (HIP start-up and the input ring's
                           // generation included), else all ranks are started again, once
  for (int i = 1; i < argc; i++)
    if (std::string(argv[i]) == "--verify")
    { // a flag without a value: take it out of the key / value pairs
      verify = true;
      for (int j = i; j + 1 < argc; j++)
        argv[j] = argv[j + 1];
      argc--;
      i--;
    }
  for (int i = 1; i + 1 < argc; i += 2)
  {
    const std::string k = argv[i];
    if (k == "--gpus") gpus = atoi(argv[i + 1]);
    else if (k == "--steps") K = atoi(argv[i + 1]);
    else if (k == "--warmup") W = atoi(argv[i + 1]);
    else if (k == "--channels") C = unsigned(atoi(argv[i + 1]));
    else if (k == "--watchdog") watchdog = unsigned(atoi(argv[i + 1]));
    else if (k == "--up-timeout") up_timeout = atoi(argv[i + 1]);
    else if (k == "--gather-root") rotate = std::string(argv[i + 1]) == "rotate";
    else if (k == "--own-stream") own_stream = atoi(argv[i + 1]);
  }
  const std::string idfile = "/tmp/fmd_node_bench_" + std::to_string(getpid()) + ".id";
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); // dmabuf IPC: what RCCL needs on this pool
  // The ranks are forked before anything touches HIP; when not every one of them has its communicator up after
  // `up_timeout` seconds (a stalled RCCL bootstrap: ~1 launch in 20 on this pool), all of them are ended and started
  // again, once, with a fresh id file (tools/rank_supervisor.hpp).
  int attempts = 0;
  const int worst = fmd_launch::run_ranks(
      gpus, up_timeout, watchdog,
      [&](int r, int attempt, int up_fd) {
        return rank_main(r, gpus, K, W, C, idfile + "." + std::to_string(attempt), verify, rotate, own_stream, up_fd);
      },
      &attempts);
  for (int a = 0; a < attempts; a++)
  {
    unlink((idfile + "." + std::to_string(a)).c_str());
    unlink((idfile + "." + std::to_string(a) + ".tmp").c_str());
  }
  return worst;
}
