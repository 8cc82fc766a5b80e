/*
 * fmd.h -- C ABI of the MI355X-native FM broadcast decoder (libfmd_hip.so).
 *
 * Drop-in boundary for ONE path of AlwinEsch/pvr.rtl.radiofm: cFmDecoder::ProcessStream()
 * and the upward RDS callbacks.  The reference has no C ABI for this path (cFmDecoder is a
 * hidden C++ class, src/FmDecode.h:91); the entry points below are what a binding of that
 * class would need, one per reference member, plus batched variants (many independent
 * channels per call) which are what the GPU is for.  include/fm_decoder.hpp puts the
 * reference's exact class surface on top of this ABI.
 *
 * All citations are relative to /root/reference/src/.  Plain pointers and sizes only; no
 * torch / HIP types (streams are passed as void* = hipStream_t).  Every function returns
 * FMD_OK or a negative error; fmd_last_error() gives the text.  There is no CPU fallback:
 * if no HIP device is usable the create calls fail.
 */
#ifndef FMD_H
#define FMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_OK 0
#define FMD_ERR_ARG (-1)     /* invalid argument / unsupported configuration */
#define FMD_ERR_DEVICE (-2)  /* HIP runtime error or no usable device */
#define FMD_ERR_SIZE (-3)    /* samples outside [fmd_batch_min_samples(), FMD_MAX_BLOCK] */
#define FMD_ERR_STATE (-4)
/* Not an error (positive): RDS groups were lost since the last report -- a call's group queue or the
 * record buffer of fmd_batch_export_rds_device was full.  Audio and channel state are intact and the
 * batch stays usable.  Returned once by fmd_batch_wait[_lagged] / fmd_batch_export_rds_device (or
 * queried with fmd_batch_take_rds_lost), then cleared.  The reference has no such condition: its
 * group decoder runs inside ProcessStream (RDSProcess.cpp:312,355), nothing is ever queued. */
#define FMD_WARN_RDS_LOST 1

/* cRtlSdrSource::default_block_length (RTL_SDR_Source.h:25): the reference's internal buffers
 * are hard-sized to it (FmDecode.cpp:277-282), so samples <= 65536 is its precondition too. */
#define FMD_MAX_BLOCK 65536u
/* Calls of at least this size are taken by every geometry.  The exact lower bound of a batch is
 * fmd_batch_min_samples() and is far smaller (88 samples at 2.4 MS/s / downsample 11): short blocks
 * are decoded the way the reference decodes them -- a half-band stage with fewer than L inputs
 * passes them on unfiltered (DownConvert.cpp:519-520), with fewer than 2 (L - 1) it refills its delay
 * line from its own outputs (:546-547, the in-place array), a block shorter than a filter keeps part
 * of the old history (:137-145, :236-253).  Refused (FMD_ERR_SIZE) are only blocks so short that some
 * stage would get no sample at all -- the reference's level meters divide by zero there
 * (FmDecode.cpp:522-539) -- or fewer than the 20 inputs its unrolled 11-tap stage reads
 * unconditionally (:596-661). */
#define FMD_MIN_BLOCK 8192u
/* A further limit inherited from the reference: samples / downsample (the baseband length of a
 * call) must stay below 32768 - 51, the size of its half-band delay lines (DownConvert.cpp:267,
 * :500; it overruns them silently).  Only matters for downsample < 3. */

/* Constructor arguments of cFmDecoder (FmDecode.h:110-116).  table_size / if_filter_order are
 * the two internal constants BASELINE configs 3 and 5 override; 0 selects the reference
 * values 64 (FmDecode.cpp:249) and 8*downsample (FmDecode.cpp:262). */
#define FMD_FIR_SEQUENTIAL 0
#define FMD_FIR_SHUFFLE_PARITY_WAIVED 0x101
#define FMD_FIR_FMA_PARITY_WAIVED 0x102
typedef struct fmd_params
{
  double sample_rate_if;
  double tuning_offset;
  double sample_rate_pcm;
  double bandwidth_pcm;
  unsigned downsample;
  int us_version;
  unsigned table_size;
  unsigned if_filter_order;
  /* How the IF FIR adds up an output's taps.  FMD_FIR_SEQUENTIAL (0, the default and the only mode
   * under the parity contract): one lane per output, taps in the reference's order
   * (DownConvert.cpp:117-121) -- bit-identical results.  FMD_FIR_SHUFFLE_PARITY_WAIVED: the sum split
   * over four lanes and combined with wavefront shuffles (the reduction BASELINE's north star names):
   * a different order of float additions.  Measured on BASELINE config 2 (6 s): audio 1.2e-5 RMS from
   * the reference (worst block 3.5e-5) -- ABOVE the 1e-5 RMS the contract allows -- and 3x slower, so
   * whoever asks for it says in the value itself that parity is waived; a plain 1 is refused.
   * Headline window layout only (odd downsample, power-of-two tuner table), other geometries ignore it.
   * FMD_FIR_FMA_PARITY_WAIVED: the reference's tap order, every multiply-add fused (v_pk_fma_f32: one rounding
   * per tap instead of two) in the IF FIR and in the two fractional resamplers (DownConvert.cpp:117-121,
   * 203-232) -- what an x86 build of the reference with -march=native does to the same loops (BASELINE.md
   * section 2: "output bits change").  Not bit-identical, so it too has to be asked for by name (a plain 2 is
   * refused); it exists to put a price on bit-exactness (docs/MEASUREMENTS.md: audio RMS distance, joules per
   * call, MS/s).  Reference geometry only (88 taps, downsample 11); refused elsewhere. */
  int fir_reduction;
} fmd_params;

/* Upward callbacks = the three cRadioReceiver members the RDS group decoder calls
 * (RadioReceiver.h:77,80,115; called from RDSGroupDecoder.cpp:403,405,981,990).  Invoked on
 * the calling thread from inside fmd_process_stream / fmd_batch_process_host.  A NULL entry
 * behaves like the reference with no dialog open (frames accepted, name accepted, inactive).
 * frame = ADD(2) SQC MFL payload CRC16(2), unstuffed, valid only during the call. */
typedef struct fmd_callbacks
{
  int (*add_uecp_frame)(void* user, unsigned channel, const uint8_t* frame, unsigned len);
  int (*set_channel_name)(void* user, unsigned channel, const char name[9]);
  int (*is_setting_active)(void* user, unsigned channel);
} fmd_callbacks;

/* Getters of cFmDecoder (FmDecode.h:140-165) */
typedef struct fmd_status
{
  int stereo_detected;   /* StereoDetected()    */
  float tuning_offset;   /* GetTuningOffset()   */
  float interface_level; /* GetInterfaceLevel() */
  float baseband_level;  /* GetBasebandLevel()  */
  float pilot_level;     /* GetPilotLevel()     */
  int rds_state;         /* 0 bit sync, 1 block sync, 2 group decode, 3 group resync */
} fmd_status;

/* One RDS group = the uint16_t[4] the signal processor hands to the group decoder
 * (RDSProcess.cpp:312,355): the bit-exact parity checkpoint. */
typedef struct fmd_rds_group
{
  uint32_t channel;
  uint32_t call_index; /* 1-based index of the process call that completed the group */
  uint16_t blocks[4];
} fmd_rds_group;

/* ---- single decoder: the cFmDecoder surface --------------------------------------- */
typedef struct fmd_decoder fmd_decoder;

/* cFmDecoder::cFmDecoder (FmDecode.cpp:237-314) */
int fmd_create(const fmd_params* params, const fmd_callbacks* cb, void* user, fmd_decoder** out);
/* cFmDecoder::~cFmDecoder (FmDecode.cpp:316-324) */
void fmd_destroy(fmd_decoder* d);
/* cFmDecoder::Reset (FmDecode.cpp:326-338) */
int fmd_reset(fmd_decoder* d);
/* cFmDecoder::ProcessStream (FmDecode.cpp:417-502): iq = samples complex<float> (host),
 * audio = caller buffer of samples*2 floats (RadioReceiver.cpp:519-520); returns the number
 * of floats written (2 per audio frame) or a negative error. */
int fmd_process_stream(fmd_decoder* d, const float* iq, unsigned samples, float* audio);
/* cRtlSdrSource::ReadAsyncCB (RTL_SDR_Source.cpp:196-213) + ProcessStream in one call: buf =
 * 2*samples bytes as librtlsdr delivers them (I, Q, I, Q, ...).  Every byte is converted with the
 * reference's float(b / (255.0 / 2.0) - 1.0) inside the IF kernel, so the result equals
 * fmd_process_stream on the converted block; the transfer and the HBM read are 4x smaller. */
int fmd_process_stream_u8(fmd_decoder* d, const uint8_t* buf, unsigned samples, float* audio);
int fmd_get_status(fmd_decoder* d, fmd_status* st);
/* The one-channel batch behind a decoder: for the profiling / development calls below (fmd_batch_set_
 * profiling, fmd_batch_get_stage_ms, fmd_batch_debug_*); not for processing (the decoder owns it). */
struct fmd_batch;
struct fmd_batch* fmd_decoder_batch(fmd_decoder* d);

/* ---- batch of independent channels on one GPU -------------------------------------- */
typedef struct fmd_batch fmd_batch;

/* All channels share params (same geometry); tuning_shifts (optional, n_channels entries)
 * overrides the cFineTuner shift per channel (config 3: many stations from one capture),
 * NULL derives it from params->tuning_offset like FmDecode.cpp:250.  device = HIP ordinal. */
int fmd_batch_create(const fmd_params* params, unsigned n_channels, const int* tuning_shifts,
                     int device, const fmd_callbacks* cb, void* user, fmd_batch** out);
void fmd_batch_destroy(fmd_batch* b);
int fmd_batch_reset(fmd_batch* b);

/* upper bounds for sizing caller buffers */
unsigned fmd_batch_channels(const fmd_batch* b);
/* How many of the batch's internal streams share a hardware queue with another stream of the process
 * (found by a probe when the batch was created; 0 = every chain of a call can overlap the others as measured).
 * HIP maps streams onto GPU_MAX_HW_QUEUES queues -- 4 unless the host process sets that variable before the
 * runtime initialises; the library reads no environment variable.  When the number is not 0, fmd_batch_create
 * still returns FMD_OK and leaves a sentence saying so in fmd_last_error(). */
int fmd_batch_streams_sharing_queue(const fmd_batch* b);
/* smallest `samples` a process call of this batch accepts (see FMD_MIN_BLOCK) */
unsigned fmd_batch_min_samples(const fmd_batch* b);
unsigned fmd_batch_max_audio_floats(const fmd_batch* b, unsigned samples);

/* Device-resident call, asynchronous on `stream` (hipStream_t, NULL = default stream).
 *  d_iq            complex<float> IQ in HBM; channel c starts at d_iq + 2*c*iq_channel_stride
 *                  floats; iq_channel_stride == 0 means one shared capture for all channels.
 *  d_audio         channel c's interleaved L/R floats at d_audio + c*audio_channel_stride.
 *  out_floats      (host, optional) floats written per channel -- the same for every
 *                  channel of a batch, known when the call returns.
 * RDS groups produced by the call stay queued on the device until fmd_batch_collect_rds /
 * fmd_batch_export_rds_device drains them.  A call appends to one of 8 queues in rotation (call index
 * mod 8), each holding max(4096, 8 x channels) groups; a caller that never drains loses the groups
 * beyond that (FMD_WARN_RDS_LOST) and nothing else. */
int fmd_batch_process_device(fmd_batch* b, const float* d_iq, size_t iq_channel_stride,
                             unsigned samples, float* d_audio, size_t audio_channel_stride,
                             unsigned* out_floats, void* stream);

/* Same with RTL-SDR byte pairs as input (see fmd_process_stream_u8): channel c starts at
 * d_iq_u8 + 2*c*iq_channel_stride bytes.  Both entry points need the pointer and the channel
 * stride to be multiples of two IQ samples (16 bytes of float IQ, 4 bytes of byte IQ). */
int fmd_batch_process_device_u8(fmd_batch* b, const uint8_t* d_iq_u8, size_t iq_channel_stride,
                                unsigned samples, float* d_audio, size_t audio_channel_stride,
                                unsigned* out_floats, void* stream);

/* Host-buffer call: copies in, runs fmd_batch_process_device, copies audio out, collects RDS
 * groups and runs the UECP group decoder (callbacks fire here).  Synchronous.  Returns FMD_OK, a
 * negative error, or FMD_WARN_RDS_LOST (once) when groups were dropped because a queue was full:
 * audio and channel state are intact. */
int fmd_batch_process_host(fmd_batch* b, const float* iq, size_t iq_channel_stride,
                           unsigned samples, float* audio, size_t audio_channel_stride,
                           unsigned* out_floats);
int fmd_batch_process_host_u8(fmd_batch* b, const uint8_t* iq_u8, size_t iq_channel_stride,
                              unsigned samples, float* audio, size_t audio_channel_stride,
                              unsigned* out_floats);

/* Copies the queued RDS groups (all channels, call order) to `out`, waits for `stream`.
 * Returns the number of groups (<= cap) or a negative error.  When run_group_decoder != 0
 * each group is also fed to that channel's UECP group decoder (callbacks fire).  The return value is
 * a count, so a loss of groups is not reported here: the flag stays for fmd_batch_wait /
 * fmd_batch_take_rds_lost / fmd_batch_process_host. */
int fmd_batch_collect_rds(fmd_batch* b, fmd_rds_group* out, unsigned cap, int run_group_decoder,
                          void* stream);

/* Like fmd_batch_collect_rds, but the `lag` (0..4) newest calls are left alone: only calls at
 * least that old are waited for and their groups drained (use with concurrency 2, where the
 * newest calls are still running). */
int fmd_batch_collect_rds_lagged(fmd_batch* b, fmd_rds_group* out, unsigned cap,
                                 int run_group_decoder, int lag, void* stream);

/* The same drain without a host round trip, for outputs that travel on as device memory (the rank-0
 * gather of a multi-GPU job): the groups of every call at least `lag` calls old are written to
 * d_records (device memory, 16-byte aligned, cap rows of 4 x int32: channel + 1 + channel_offset,
 * call_index, blocks[0] | blocks[1] << 16, blocks[2] | blocks[3] << 16; rows beyond the groups found
 * are zero, so a fixed-size message can be sent as is) by a kernel on `stream`, and those queues
 * are emptied.  Asynchronous; rows are in no particular order.  More groups than `cap` rows: the
 * surplus is lost and FMD_WARN_RDS_LOST is reported once (by a later wait / export call); the batch
 * stays usable.  Use either this or fmd_batch_collect_rds on a batch, not both for the same calls.
 * One stream at a time: concurrent exports of one batch on different streams are not supported. */
int fmd_batch_export_rds_device(fmd_batch* b, int32_t* d_records, unsigned cap, unsigned channel_offset,
                                int lag, void* stream);

/* Several captures in one batch (BASELINE configs[2] scaled out: G captures x k stations each, where config 3
 * as written is one capture x 256): channels [g k, (g + 1) k) all tune capture g -- the only stage that sees the
 * capture is the tuner in front of the IF filter (cFineTuner::Process, FmDecode.cpp:66-82); everything behind it is
 * per channel as ever.  With k > 1 the iq_channel_stride of the process calls is the distance between CAPTURES
 * (G = channels / k input rows instead of one per channel); k = 0 / 1 restores one row per channel.  iq_channel_stride
 * == 0 still means a single capture for the whole batch.  Not while calls are in flight (the device is drained). */
int fmd_batch_set_channels_per_capture(fmd_batch* b, unsigned channels_per_capture);

/* Internal execution.  A call is four independent kernel chains (FIR -> serial demodulator ->
 * {RDS branch, audio branch}); mode selects where they run:
 *   0  all on the caller's stream, in order
 *   1  on internal streams, the caller's stream is ordered after each call (default; same
 *      observable semantics as 0)
 *   2  on internal streams and the caller's stream is NOT ordered after the call: the FIR of
 *      call k+1 overlaps the serial stages of call k.  The caller must use different audio buffers
 *      for calls in flight, keep d_iq and d_audio valid, and call fmd_batch_wait[_lagged] (or
 *      collect_rds) before consuming outputs. */
int fmd_batch_set_concurrency(fmd_batch* b, int mode);
/* Orders `stream` after every call submitted so far (outputs complete, inputs released). */
int fmd_batch_wait(fmd_batch* b, void* stream);
/* lag = 1..4: every call except the newest `lag` ones (whose kernels may still be running: a call is
 * complete about 1.3 periods after its serial stage has started, so a host that must never block
 * consumes outputs two to three calls late). */
int fmd_batch_wait_lagged(fmd_batch* b, int lag, void* stream);
/* 1 if RDS groups were lost since the last report (see FMD_WARN_RDS_LOST; clears the flag), else 0. */
int fmd_batch_take_rds_lost(fmd_batch* b);

/* The getters of cFmDecoder for one channel (FmDecode.h:140-165).  They return the status the
 * newest COMPLETED call left behind (all zero before the first call; Reset zeroes what
 * cFmDecoder::Reset zeroes): the last kernel of every call writes a small record per channel into
 * host-mapped memory, and the getters only read that record -- no device synchronisation, no stream
 * operation, nothing of the batch is modified.  They may be called from any thread at any time,
 * also while another thread is inside a process call on the same batch (Kodi's status thread does
 * that: RadioReceiver.cpp:544-572 against :524).  A record is always one call's values, never a mix
 * of two writes (rds_state, which is no cFmDecoder getter, is a word of its own that the bit recovery
 * updates).  With overlapped calls (concurrency 2) the interface / baseband meters in it may already
 * include the following call. */
int fmd_batch_get_status(fmd_batch* b, unsigned channel, fmd_status* st);
/* index (1-based) of the call whose status the getters return at this moment, 0 = none yet */
int fmd_batch_status_call_index(fmd_batch* b, unsigned channel, uint32_t* call_index);

/* cRadioReceiver's audio level meter over the audio a call produced (RadioReceiver.cpp:526-528,
 * SamplesMeanRMS :584-598): float sums over the interleaved samples of the packet, then
 * level = 0.95 * level + 0.05 * rms.  Computed on the device while the audio is written; `level`
 * starts at 0 when the batch is created and, like m_AudioLevel, is not touched by Reset. */
typedef struct fmd_audio_level
{
  float mean;  /* audio_mean of the last call */
  float rms;   /* audio_rms of the last call  */
  float level; /* m_AudioLevel                */
} fmd_audio_level;
int fmd_batch_get_audio_level(fmd_batch* b, unsigned channel, fmd_audio_level* out);

/* Stage taps for parity tests: copies stage output of the last call for one channel to host.
 * Returns element count (complex counts as one) or negative error. */
enum fmd_tap
{
  FMD_TAP_DEMOD = 0,    /* complex: IF FIR output            */
  FMD_TAP_BASEBAND = 1, /* FM PLL output                     */
  FMD_TAP_PILOT38 = 2,  /* 38 kHz * 2 * baseband             */
  FMD_TAP_MONO_RS = 3,  /* mono resampler output             */
  FMD_TAP_STEREO_RS = 4,/* stereo resampler output           */
  FMD_TAP_RDS_LPF = 5,  /* complex: RDS 75-tap LPF output    */
  FMD_TAP_RDS_PLL = 6,
  FMD_TAP_RDS_MF = 7,
  FMD_TAP_RDS_SYNC = 8
};
int fmd_batch_get_tap(fmd_batch* b, int tap, unsigned channel, float* out, unsigned cap_floats);
/* The three RDS-recurrence taps (PLL, matched filter, bit sync) cost extra stores per sample and
 * are only written while enabled (default off). */
int fmd_batch_set_debug_taps(fmd_batch* b, int enable);

/* Design constants / taps as the host computed them (for parity with the oracle). */
int fmd_batch_get_design(fmd_batch* b, int what, float* out, unsigned cap);
enum fmd_design_item
{
  FMD_DESIGN_IF_TAPS = 0,
  FMD_DESIGN_RS_TAPS = 1,
  FMD_DESIGN_AUDIO_LPF = 2,
  FMD_DESIGN_RDS_LPF = 3,
  FMD_DESIGN_RDS_MF = 4,
  FMD_DESIGN_SCALARS = 5, /* same order as oracle fmo_get_constants */
  FMD_DESIGN_LUT0 = 6     /* channel 0 cFineTuner table, interleaved */
};

/* Device time per stage, measured with HIP events recorded on the call's own stream.
 * level 0 = off, 1 = events around the IF FIR kernel only, 2 = around every stage.  Each call
 * made while profiling is on gets its own event set (no synchronisation inside the calls);
 * fmd_batch_get_stage_ms synchronises the device, writes the AVERAGE ms per stage over those
 * calls (-1 for stages not covered at level 1; index = fmd_stage_name index) and returns the
 * number of calls averaged.  set_profiling restarts the averaging window. */
int fmd_batch_set_profiling(fmd_batch* b, int level);
int fmd_batch_get_stage_ms(fmd_batch* b, float* out, unsigned cap);
const char* fmd_stage_name(unsigned idx);

/* Test aid: evaluates the device build of one math helper of csrc/fmd_math.h on n arguments
 * (host arrays).  what: 0 atan2f table form (a = y, b = x), 1 atan2f literal fdlibm, 2 sin/cos
 * table form (a = phase; out0 = sin, out1 = cos), 3 sin/cos series form, 4 mid-range division
 * a / b, 5 RTL-SDR byte -> float (a = byte value), 6 the RDS PLL's polynomial arctan2, 7 sin/cos of a
 * phase in [0, 8) with the exact float reduction (the serial stage's two NCOs). */
int fmd_debug_math(int what, unsigned n, const float* a, const float* b, float* out0, float* out1);

/* Dev aid, only with FMD_SERIAL_PROBE=1 in the environment at batch creation: per workgroup of the
 * serial stage's last 8 launches, (start, end) on the device's 100 MHz clock and the shader-clock
 * cycles in between (low 40 bits; HW_ID bits 8-15 and XCC_ID above): 8 x (padded channels / 64) records of 3 x int64, launch = call index mod 8,
 * records of workgroups that did not exist stay zero.  Returns the number of records written (0
 * when the probe is off).  Synchronises the device. */
int fmd_batch_debug_serial_probe(fmd_batch* b, long long* out, unsigned cap_workgroups);
/* Dev aid: which internal streams share a hardware queue with the caller's `stream` (bit i: internal stream i waits
 * behind it; bit 8 + i: it waits behind internal stream i; 0 = none).  Drains the device. */
int fmd_batch_debug_stream_conflicts(fmd_batch* b, void* stream);
/* Dev aid (overlapped calls at profiling level 1, whole-CU serial stage): per profiled call when its
 * IF FIR, its serial stage, its audio tail, its half-band chain and its resampler started and ended on
 * the device (the last two only in their large-batch forms), in ms since the first profiled call's FIR
 * started: cap_calls rows of 10 floats (-1 = not recorded).  Returns the number of rows.  Synchronises
 * the device.  What a short run's fill and drain are made of (bench.py prints it with
 * FMD_BENCH_TIMELINE=1). */
int fmd_batch_debug_timeline(fmd_batch* b, float* out, unsigned cap_calls);
/* Test aid: bound (in polls) of the serial stage's LDS hand-off waits for the calls that follow;
 * 0 makes every wait time out at once, which exercises the device-side error path. */
int fmd_batch_debug_set_spin_limit(fmd_batch* b, unsigned limit);
/* Development switches of one batch, by name (the shipped library reads NO environment variable).
 * They select between implementations that give identical results, or add instrumentation; they
 * apply to the calls that follow and may be changed while no call is in flight.  Returns FMD_ERR_ARG
 * for an unknown key.  Keys: see fmd_batch_debug_set in csrc/fmd_batch.hip ("resampler": -1 the
 * library decides, 0 window per wave (k_resample), 1 LDS ring (k_resample_ring) wherever the
 * geometry allows it; ...). */
int fmd_batch_debug_set(fmd_batch* b, const char* key, int value);
/* Where the time of the host-buffer calls (fmd_batch_process_host*, fmd_process_stream*) went since the
 * last query, mean ms per call: out[0] copy of the IQ block to the device, [1] submission of the call's
 * kernels, [2] waiting for them + copy of the audio back, [3] collection of the RDS groups + UECP group
 * decoder callbacks.  Returns the number of calls averaged. */
int fmd_batch_debug_host_ms(fmd_batch* b, float out[4]);

const char* fmd_last_error(void);
const char* fmd_version(void);

/* ---- the stream side of cRadioReceiver around the decoder --------------------------- */
/* What turns blocks of IQ into Kodi demux packets and the signal-status maths on
 * top of the decoder's getters (8(f)-4): cRadioReceiver::OpenLiveStream's stream state
 * (RadioReceiver.cpp:296-349), WriteDataBuffer / EndDataBuffer / SourceGetSamples (:426-460),
 * AddUECPDataFrame (:387-414), DemuxRead (:462-542) and both GetSignalStatus (:544-582).  The
 * decoder inside is an fmd_decoder; the rest is host bookkeeping like the reference's.
 * Not thread-safe by itself except write/end against demux_read (producer / consumer, like the
 * reference's source thread and demux thread). */
typedef struct fmd_receiver fmd_receiver;

#define FMD_STREAM_AUDIO 1          /* PID 1, pcm_f32le 2 ch 48 kHz (RadioReceiver.cpp:308-316) */
#define FMD_STREAM_RDS 2            /* PID 2, rds: byte-stuffed UECP frames (:326-334)          */
#define FMD_STREAM_CHANGE (-11)     /* DEMUX_SPECIALID_STREAMCHANGE                             */
#define FMD_STREAM_TIME_BASE 1000000 /* Kodi STREAM_TIME_BASE (microseconds)                    */

/* the DEMUX_PACKET fields DemuxRead fills */
typedef struct fmd_demux_packet
{
  int stream_id;       /* iStreamId                                                  */
  int size;            /* iSize, bytes                                               */
  double pts;          /* pts                                                        */
  double duration;     /* duration (audio packets only, else 0)                      */
  const uint8_t* data; /* pData: valid until the next call on this receiver          */
} fmd_demux_packet;

/* OpenLiveStream's decoder + stream state: decoder for (params), m_StreamChange = true,
 * m_PTSNext = STREAM_TIME_BASE, Reset().  tuner_freq = m_activeTunerFreq (Hz), adapter_name =
 * the RTL-SDR device name (both only appear in the status text). */
int fmd_receiver_open(const fmd_params* params, double tuner_freq, const char* adapter_name,
                      fmd_receiver** out);
void fmd_receiver_close(fmd_receiver* r);
/* WriteDataBuffer (:426-436): queues one block (copied).  _u8: cRtlSdrSource::ReadAsyncCB's input,
 * converted inside the IF kernel when the block is decoded. */
int fmd_receiver_write_iq(fmd_receiver* r, const float* iq, unsigned samples);
int fmd_receiver_write_u8(fmd_receiver* r, const uint8_t* buf, unsigned samples);
void fmd_receiver_end(fmd_receiver* r);                 /* EndDataBuffer (:438-443)        */
size_t fmd_receiver_queued_samples(fmd_receiver* r);    /* SourceQueuedSamples (:420-424)  */
void fmd_receiver_set_stream_change(fmd_receiver* r);   /* SetStreamChange (RadioReceiver.h:83) */
/* DemuxRead (:462-542): 1 = packet filled, 0 = no packet (the reference returns nullptr: end
 * marked and queue empty), negative = error.  Order per call like the reference: stream-change
 * packet, else pending RDS bytes, else decode the next IQ block into an audio packet.  Blocks
 * while the queue is empty and the end is not marked (polling every 20 ms like :448). */
int fmd_receiver_demux_read(fmd_receiver* r, fmd_demux_packet* pkt);
/* GetSignalStatus(float&, float&, bool&) (:544-556): 1 = values valid, 0 = no decoder / stream
 * change pending (the reference returns false). */
int fmd_receiver_signal_status(fmd_receiver* r, float* interface_level_db, float* audio_level_db,
                               int* stereo);
/* GetSignalStatus(int, PVRSignalStatus&) (:558-582) */
typedef struct fmd_pvr_signal_status
{
  char adapter_name[128];
  char adapter_status[256];
  char provider_name[64]; /* m_channelName (trimmed PS name) */
  int signal;             /* SetSignal(2.5 * (interfaceLevel + 40) * 656) */
  int snr;                /* SetSNR((audioLevel + 100) * 656)             */
} fmd_pvr_signal_status;
int fmd_receiver_pvr_signal_status(fmd_receiver* r, fmd_pvr_signal_status* out);
/* the decoder inside (status getters, not to be destroyed) */
fmd_decoder* fmd_receiver_decoder(fmd_receiver* r);

/* ---- host-only pieces (no GPU needed) ----------------------------------------------- */
/* UECP group decoder = cRDSGroupDecoder (RDSGroupDecoder.cpp:166-1001). */
typedef struct fmd_group_decoder fmd_group_decoder;
fmd_group_decoder* fmd_group_decoder_create(const fmd_callbacks* cb, void* user, unsigned channel);
void fmd_group_decoder_destroy(fmd_group_decoder* g);
void fmd_group_decoder_reset(fmd_group_decoder* g);
void fmd_group_decoder_push(fmd_group_decoder* g, const uint16_t blocks[4]);
/* Filter design as the constructors do it on the host (float/double promotions as written):
 * cDownsampleFilter's Lanczos table (DownConvert.cpp:18-56 through the ctor :78; `order` is the
 * ctor's filter_order, order + 2 floats are written), cFirFilter::InitLPFilter's Kaiser low-pass
 * (FirFilter.cpp:44-140; returns the tap count), cIirFilter::Init (IirFilter.cpp:11-60; type
 * 0 LP, 1 HP, 2 BP, 3 BR; out = b0 b1 b2 a1 a2) and cFineTuner's table (FmDecode.cpp:45-58;
 * 2*table_size floats).  Return the element count (which may exceed cap: nothing past cap is
 * written) or -1. */
int fmd_design_lanczos(unsigned order, double cutoff, float* out, unsigned cap);
int fmd_design_lp_kaiser(float scale, float astop, float fpass, float fstop, float fs, float* out,
                         unsigned cap);
int fmd_design_biquad(int type, float f0, float q, float fs, float out[5]);
int fmd_design_tuner_lut(unsigned table_size, int freq_shift, float* out, unsigned cap);

/* cRadioReceiver::AddUECPDataFrame byte stuffing (RadioReceiver.cpp:387-414):
 * 0xFE, payload with 0xFD escapes, 0xFF.  Returns bytes written (<= cap) or -1. */
int fmd_uecp_stuff_frame(const uint8_t* frame, unsigned len, uint8_t* out, unsigned cap);

#ifdef __cplusplus
}
#endif
#endif
