/*
 * fmd_gather.h -- C ABI of the whole-node step: one process per GPU, every rank decodes its own
 * channels (include/fmd.h, no exchange during compute) and rank 0 receives every rank's float audio
 * and RDS group records once per step over RCCL (xGMI).  libfmd_gather.so = this + librccl; a
 * single-GPU user of libfmd_hip.so never loads it.
 *
 * The reference has no counterpart: cRadioReceiver decodes one station on one CPU thread
 * (/root/reference/src/RadioReceiver.cpp:515-538).  What is gathered is what its DemuxRead hands on
 * per channel: the interleaved float audio of a ProcessStream call (FmDecode.cpp:473-501) and the
 * RDS groups the signal processor found in it (RDSProcess.cpp:312,355).
 *
 * RCCL has no gather: a step is one group of ncclSend (every rank, its two buffers) / ncclRecv (rank 0,
 * two per rank), so that each peer uses its own xGMI link to rank 0.  It runs on a stream of the
 * library's own, behind an event on the caller's stream, and overlaps the next steps' compute; the
 * caller orders a stream behind it (fmd_gather_wait) before it reuses the buffers.
 */
#ifndef FMD_GATHER_H
#define FMD_GATHER_H

#include <stddef.h>
#include <stdint.h>

#include "fmd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_GATHER_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct fmd_gather fmd_gather;

/* Message of the last failed fmd_gather_* call on this thread. */
const char* fmd_gather_last_error(void);

/* Rank 0 makes the communicator's id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by
 * whatever means the job has (a file, an environment variable, MPI, a torch store). */
int fmd_gather_unique_id(uint8_t id[FMD_GATHER_ID_BYTES]);

/* Collective over all `world` ranks (ncclCommInitRank).  audio_floats / rds_rows: size of one rank's
 * message per step -- [channels][audio_stride] floats and rds_rows x 4 int32 records
 * (fmd_batch_export_rds_device's rows) -- the same on every rank. */
int fmd_gather_create(const uint8_t id[FMD_GATHER_ID_BYTES], int rank, int world, int device,
                      size_t audio_floats, unsigned rds_rows, fmd_gather** out);
void fmd_gather_destroy(fmd_gather* g);

/* One step's outputs on their way to rank 0.  The RDS groups of every call of `batch` at least `lag`
 * calls old are first drained into d_rds as records (fmd_batch_export_rds_device, channel numbers
 * offset by channel_offset, on `stream`); then, behind everything `stream` has been given so far
 * (the caller has ordered it behind the calls whose audio is in d_audio: fmd_batch_wait_lagged), the
 * library's own stream sends d_audio and d_rds; rank 0 receives rank r's into
 * d_all_audio + r * audio_floats and d_all_rds + r * rds_rows * 4 (its own by a device copy -- none where the
 * caller passes d_audio == d_all_audio / d_rds == d_all_rds, i.e. had rank 0's outputs produced in place).  Returns
 * at once; FMD_WARN_RDS_LOST like the export.  batch may be NULL (d_rds is sent as it is). */
int fmd_gather_step(fmd_gather* g, fmd_batch* batch, int lag, unsigned channel_offset, const float* d_audio,
                    int32_t* d_rds, float* d_all_audio, int32_t* d_all_rds, void* stream);

/* The same step with rank `root` as its receiver (every rank passes the same root for the same step; d_all_audio /
 * d_all_rds are needed on that rank only; its own part is rank root's slot: in place when d_audio == d_all_audio +
 * root * audio_floats).  A caller that rotates the root -- step i to rank i % world -- spreads what rank 0 alone
 * would take: 7 x 88 MB of writes per step into one GPU's memory system at 8 GPUs, which costs that GPU 8-10 % of
 * its throughput (emulated on one GPU, docs/MEASUREMENTS.md round 6) and with it the node, against ~1 % on every
 * GPU.  Each rank then holds every world-th step's outputs of all channels. */
int fmd_gather_step_root(fmd_gather* g, int root, fmd_batch* batch, int lag, unsigned channel_offset,
                         const float* d_audio, int32_t* d_rds, float* d_all_audio, int32_t* d_all_rds, void* stream);

/* Orders `stream` behind every step issued so far (before d_audio / d_rds / the receive buffers are
 * written again or read); _lagged: behind all but the `lag` (< 16) youngest -- a caller that rotates
 * its buffers waits only for the step that last used the one it is about to write. */
int fmd_gather_wait(fmd_gather* g, void* stream);
int fmd_gather_wait_lagged(fmd_gather* g, unsigned lag, void* stream);

/* All ranks meet (an all-reduce of one word on the communicator's stream, then a host wait): the
 * barrier around a timed region.  *max_value = the largest `value` over the ranks (may be NULL). */
int fmd_gather_barrier(fmd_gather* g, double value, double* max_value);

/* Mean duration of a step's send / receive group on the library's stream, in ms, over the steps issued
 * since the last call (device events); < 0 when there was none.  Synchronises that stream. */
float fmd_gather_ms_per_step(fmd_gather* g);

/* What the communicator itself says it is (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): a bench line
 * that carries ranks_seen proves RCCL connected N ranks, not that N processes each ran a world of one. */
typedef struct fmd_gather_info_t
{
  int ranks_seen;   /* ncclCommCount */
  int rank;         /* ncclCommUserRank */
  int device;       /* ncclCommCuDevice */
  int world_asked;  /* fmd_gather_create's argument */
  uint64_t steps_issued;
} fmd_gather_info_t;
int fmd_gather_info(fmd_gather* g, fmd_gather_info_t* out);

/* Measurement aid, world of ONE only (refused otherwise): every later step also WRITES, on the library's stream and
 * in the step's place, what `peers` more ranks' receives would write into rank 0's buffers -- peers x (audio_floats
 * floats + rds_rows records) at d_all_audio + r * audio_floats / d_all_rds + r * rds_rows * 4, r = 1 .. peers (the
 * caller's receive buffers must hold 1 + peers ranks) -- with `workgroups_per_peer` workgroups per peer, as RCCL's
 * receive kernel would occupy a few CUs.  What it is for: sizing rank 0's extra load (its memory system and its
 * power budget take 7 x 88 MB per step on an 8-GPU node) on a box with one GPU, before the first 8-GPU run
 * (docs/MEASUREMENTS.md).  The bytes arrive as fast as the kernel can store them, not paced by
 * seven xGMI links: a burst, i.e. the contention of a step is concentrated, its joules are the same.  0 = off. */
int fmd_gather_debug_emulate_peers(fmd_gather* g, int peers, int workgroups_per_peer);
/* ... and, on top of it, which role this one GPU plays in a node of `world` = peers + 1 ranks: `every` = 1 the root of
 * every step (rank 0 of a fixed-root gather: the default above); `every` = world a rank of a ROTATING root -- the
 * receives of `peers` ranks in every world-th step, and in the others what a sender does to its own memory system: its
 * message read once (audio_floats floats + the records, `workgroups_per_peer` workgroups); `every` = 0 a sender in
 * every step (ranks 1 .. of a fixed-root gather). */
int fmd_gather_debug_emulate_role(fmd_gather* g, int every);

#ifdef __cplusplus
}
#endif
#endif
