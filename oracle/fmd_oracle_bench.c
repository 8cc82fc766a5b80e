/*
 * fmd_oracle_bench.c -- the CPU oracle timed on native threads (bench.py's cpu_baseline leg).
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/ (see fmd_oracle.h).  What is timed is
 * fmo_process_stream, the restatement of cFmDecoder::ProcessStream (/root/reference/src/
 * FmDecode.cpp:417-502), one decoder per POSIX thread, every thread its own channel state on the
 * same read-only input blocks replayed in a loop -- the way SURVEY 8(d) asks for the whole-host
 * figure ("T threads, channels statically partitioned").  No Python, no GIL, no allocation inside
 * the timed loops.
 */
#define _GNU_SOURCE
#include "fmd_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* Start gate: every thread reports ready and waits for `go`; the caller opens the gate once all
 * threads it managed to start are ready -- with `run` = 0 when it could not start them all, so that
 * the ones that exist leave at once and can be joined (a barrier sized for the full count would keep
 * them waiting for ever). */
typedef struct
{
  pthread_mutex_t mu;
  pthread_cond_t cv;
  unsigned ready;
  int go, run;
} start_gate;

typedef struct
{
  const fmo_params* params;
  const float* blocks; /* nblocks x samples x 2 floats */
  unsigned nblocks, samples;
  double seconds;
  start_gate* start;
  unsigned long long calls; /* out */
  double elapsed;           /* out: seconds inside this thread's timed loop */
} bench_job;

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* bench_thread(void* arg)
{
  bench_job* j = (bench_job*)arg;
  fmo_decoder* d = fmo_create(j->params);
  float* audio = (float*)malloc(sizeof(float) * 2u * (size_t)j->samples); /* RadioReceiver.cpp:519-520 */
  const size_t stride = (size_t)j->samples * 2u;
  if (d && audio)
    for (unsigned b = 0; b < 2 && b < j->nblocks; b++) /* warm caches and the PLLs */
      fmo_process_stream(d, j->blocks + stride * b, j->samples, audio);
  int run;
  pthread_mutex_lock(&j->start->mu);
  j->start->ready++;
  pthread_cond_broadcast(&j->start->cv);
  while (!j->start->go)
    pthread_cond_wait(&j->start->cv, &j->start->mu);
  run = j->start->run;
  pthread_mutex_unlock(&j->start->mu); /* all threads enter their timed loops together */
  unsigned long long n = 0;
  const double t0 = now_s();
  double t = t0;
  if (d && audio && run)
    while (t - t0 < j->seconds)
    {
      fmo_process_stream(d, j->blocks + stride * (n % j->nblocks), j->samples, audio);
      n++;
      t = now_s();
    }
  j->calls = n;
  j->elapsed = t - t0;
  free(audio);
  if (d)
    fmo_destroy(d);
  return NULL;
}

/* Runs `threads` decoders for `seconds` each.  Returns the whole-host rate in IQ samples per
 * second = sum over threads of (calls x samples / that thread's elapsed time); total_calls and
 * max_elapsed (optional) describe the sample.  0 on failure. */
double fmo_bench_threads(const fmo_params* p, unsigned threads, double seconds, const float* blocks,
                         unsigned nblocks, unsigned samples, unsigned long long* total_calls,
                         double* max_elapsed)
{
  if (!p || !blocks || !threads || !nblocks || !samples || samples > FMO_MAX_BLOCK)
    return 0.0;
  bench_job* jobs = (bench_job*)calloc(threads, sizeof(bench_job));
  pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
  start_gate start;
  if (!jobs || !th || pthread_mutex_init(&start.mu, NULL) != 0)
  {
    free(jobs);
    free(th);
    return 0.0;
  }
  if (pthread_cond_init(&start.cv, NULL) != 0)
  {
    pthread_mutex_destroy(&start.mu);
    free(jobs);
    free(th);
    return 0.0;
  }
  start.ready = 0;
  start.go = 0;
  start.run = 0;
  unsigned started = 0;
  for (unsigned i = 0; i < threads; i++)
  {
    jobs[i].params = p;
    jobs[i].blocks = blocks;
    jobs[i].nblocks = nblocks;
    jobs[i].samples = samples;
    jobs[i].seconds = seconds;
    jobs[i].start = &start;
    if (pthread_create(&th[i], NULL, bench_thread, &jobs[i]) != 0)
      break;
    started++;
  }
  double rate = 0.0, worst = 0.0;
  unsigned long long calls = 0;
  /* open the gate when every thread that exists is ready; they only run if all of them exist */
  pthread_mutex_lock(&start.mu);
  while (start.ready < started)
    pthread_cond_wait(&start.cv, &start.mu);
  start.run = started == threads;
  start.go = 1;
  pthread_cond_broadcast(&start.cv);
  pthread_mutex_unlock(&start.mu);
  for (unsigned i = 0; i < started; i++)
    pthread_join(th[i], NULL);
  if (started == threads)
    for (unsigned i = 0; i < threads; i++)
    {
      if (jobs[i].elapsed > 0)
        rate += (double)jobs[i].calls * (double)samples / jobs[i].elapsed;
      calls += jobs[i].calls;
      if (jobs[i].elapsed > worst)
        worst = jobs[i].elapsed;
    }
  pthread_cond_destroy(&start.cv);
  pthread_mutex_destroy(&start.mu);
  if (total_calls)
    *total_calls = calls;
  if (max_elapsed)
    *max_elapsed = worst;
  free(jobs);
  free(th);
  return rate;
}
