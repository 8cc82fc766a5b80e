/*
 * fmd_receiver.hpp -- a demux session around one GPU decoder (fmd_receiver_* of include/fmd.h).
 *
 * What a host that is not Kodi needs between its IQ source thread and its packet consumer, with
 * the packet stream byte-identical to what cRadioReceiver hands to Kodi.  The design is this
 * library's own; only the observable behaviour follows the reference (all file:line relative to
 * /root/reference/src/):
 *   packet order per read: stream change, else pending RDS bytes, else one decoded IQ block
 *                                                                 RadioReceiver.cpp:462-542
 *   audio packet: pcm_f32le, duration = floats * STREAM_TIME_BASE / 2 / 48000, pts running sum
 *                                                                 RadioReceiver.cpp:531-539
 *   RDS packet: 0xFE, frame with 0xFD escapes, 0xFF per UECP frame; a frame is refused while more
 *   than 16384 bytes are pending                                  RadioReceiver.cpp:387-414
 *   both GetSignalStatus: dB values, Signal / SNR integers, status text
 *                                                                 RadioReceiver.cpp:544-582
 *   PS name trimmed of white space                                RadioReceiver.cpp:600-612
 *
 * Three parts:
 *   BlockPool  IQ blocks in page-locked host memory, recycled through a free list: the H2D copy of
 *              a block is a DMA from pinned memory, and a steady stream allocates nothing.
 *   UecpBytes  the byte-stuffed RDS stream waiting for its packet.
 *   Receiver   packet sequencing, the presentation clock, status read-outs.
 * The audio level meter of the reference's demux loop (SamplesMeanRMS over the packet + EMA,
 * RadioReceiver.cpp:526-528) is computed on the device together with the audio (k_audio_tail) and
 * read back through fmd_batch_get_audio_level.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/fmd.h"

namespace fmd
{

/* 0xFE, payload with 0xFD / 0xFE / 0xFF written as 0xFD followed by (v & 3) - 1, 0xFF
 * (RadioReceiver.cpp:387-414).  emit(byte) receives the stuffed stream. */
template <class Emit>
inline void uecp_stuff(const uint8_t* frame, unsigned len, Emit&& emit)
{
  emit(uint8_t(0xFE));
  for (const uint8_t* p = frame; p != frame + len; ++p)
  {
    if (*p >= 0xFD)
    {
      emit(uint8_t(0xFD));
      emit(uint8_t((*p & 3u) - 1u));
    }
    else
      emit(*p);
  }
  emit(uint8_t(0xFF));
}

/* One block of IQ as the source delivered it, in page-locked host memory (pageable once the pool's
 * page-locked budget is used up: the copy to the device still works, only slower). */
struct IqBlock
{
  void* mem = nullptr;  // hipHostMalloc, or malloc when !pinned
  bool pinned = false;
  size_t capacity = 0;  // bytes
  unsigned samples = 0;
  bool bytes_in = false; // RTL-SDR (I, Q) byte pairs instead of complex<float>
  IqBlock* next = nullptr;
};

/* FIFO of filled blocks plus a free list of empty ones.  One producer (the source thread) and one
 * consumer (the demux thread); the queue is unbounded like the reference's (it warns about a
 * growing backlog, it never drops), but blocks are reused once the consumer hands them back.
 * Blocks come in power-of-two size classes from 16 KiB (a 16 k-sample byte block is 32 KiB, a full
 * float block 512 KiB), and at most kPinnedBudget bytes of them are page-locked: a consumer that falls
 * behind makes the backlog grow in ordinary memory, not in the machine's pinned pages. */
class BlockPool
{
public:
  static constexpr size_t kPinnedBudget = size_t(64) << 20;

  ~BlockPool()
  {
    release_chain(m_head);
    release_chain(m_free);
  }

  /* a few blocks up front, from the thread that opens the stream (page-locking takes milliseconds:
   * not something the source thread should do in the middle of a delivery) */
  void prime(unsigned blocks, size_t bytes)
  { // all of them taken first, then given back: a block recycled at once would be the next one taken
    std::vector<IqBlock*> got;
    for (unsigned i = 0; i < blocks; i++)
      if (IqBlock* blk = take_free(bytes))
        got.push_back(blk);
    for (IqBlock* blk : got)
      recycle(blk);
  }

  /* copies `bytes` bytes into a recycled (or new) pinned block and appends it; false = out of
   * memory */
  bool push(const void* data, size_t bytes, unsigned samples, bool bytes_in)
  {
    IqBlock* blk = take_free(bytes);
    if (!blk)
      return false;
    std::memcpy(blk->mem, data, bytes);
    blk->samples = samples;
    blk->bytes_in = bytes_in;
    blk->next = nullptr;
    {
      std::lock_guard<std::mutex> g(m_lock);
      if (m_tail)
        m_tail->next = blk;
      else
        m_head = blk;
      m_tail = blk;
      m_backlog += samples;
    }
    m_wake.notify_one();
    return true;
  }

  void close()
  {
    {
      std::lock_guard<std::mutex> g(m_lock);
      m_closed = true;
    }
    m_wake.notify_all();
  }

  /* next block in arrival order; waits while the queue is empty and the source has not ended.
   * nullptr = ended and drained. */
  IqBlock* pop()
  {
    std::unique_lock<std::mutex> g(m_lock);
    // bounded waits (20 ms, the reference's poll interval): a lost wake-up costs one interval at most
    while (!m_head && !m_closed)
      m_wake.wait_for(g, std::chrono::milliseconds(20));
    IqBlock* blk = m_head;
    if (blk)
    {
      m_head = blk->next;
      if (!m_head)
        m_tail = nullptr;
      m_backlog -= blk->samples;
      blk->next = nullptr;
    }
    return blk;
  }

  void recycle(IqBlock* blk)
  {
    std::lock_guard<std::mutex> g(m_lock);
    blk->next = m_free;
    m_free = blk;
  }

  size_t backlog()
  {
    std::lock_guard<std::mutex> g(m_lock);
    return m_backlog;
  }

private:
  static size_t size_class(size_t bytes)
  {
    size_t c = size_t(16) << 10;
    while (c < bytes)
      c <<= 1;
    return c;
  }
  IqBlock* take_free(size_t bytes)
  {
    IqBlock* blk = nullptr;
    {
      std::lock_guard<std::mutex> g(m_lock);
      // first free block that is large enough (the list holds few blocks, mostly of one class)
      for (IqBlock** pp = &m_free; *pp; pp = &(*pp)->next)
        if ((*pp)->capacity >= bytes)
        {
          blk = *pp;
          *pp = blk->next;
          break;
        }
    }
    if (blk)
      return blk;
    blk = new (std::nothrow) IqBlock;
    if (!blk)
      return nullptr;
    const size_t want = size_class(bytes);
    bool pin;
    {
      std::lock_guard<std::mutex> g(m_lock);
      pin = m_pinned_bytes + want <= kPinnedBudget;
      if (pin)
        m_pinned_bytes += want;
    }
    // portable: valid for every device and from every thread (the source thread never selects one)
    if (pin && hipHostMalloc(&blk->mem, want, hipHostMallocPortable) == hipSuccess)
      blk->pinned = true;
    else
    {
      if (pin)
      {
        std::lock_guard<std::mutex> g(m_lock);
        m_pinned_bytes -= want;
      }
      blk->mem = std::malloc(want);
      blk->pinned = false;
    }
    if (!blk->mem)
    {
      delete blk;
      return nullptr;
    }
    blk->capacity = want;
    return blk;
  }
  static void release_chain(IqBlock* b)
  {
    while (b)
    {
      IqBlock* n = b->next;
      if (b->mem && b->pinned)
        (void)hipHostFree(b->mem);
      else if (b->mem)
        std::free(b->mem);
      delete b;
      b = n;
    }
  }

  std::mutex m_lock;
  std::condition_variable m_wake;
  IqBlock* m_head = nullptr;
  IqBlock* m_tail = nullptr;
  IqBlock* m_free = nullptr;
  size_t m_backlog = 0; // samples queued
  size_t m_pinned_bytes = 0;
  bool m_closed = false;
};

/* The RDS byte stream between the group decoder (which appends whole frames from inside the
 * decode call) and the packet reader (which takes everything that is pending). */
class UecpBytes
{
public:
  static constexpr size_t kRefuseAbove = 16384; // RadioReceiver.cpp:389

  bool append_frame(const uint8_t* frame, unsigned len)
  {
    std::lock_guard<std::mutex> g(m_lock);
    if (m_pending.size() > kRefuseAbove)
      return false; // the consumer is not keeping up: the frame is dropped, as in the reference
    uecp_stuff(frame, len, [this](uint8_t v) { m_pending.push_back(v); });
    return true;
  }
  /* moves the pending bytes into `out`; false = nothing pending */
  bool take_all(std::vector<uint8_t>& out)
  {
    std::lock_guard<std::mutex> g(m_lock);
    if (m_pending.empty())
      return false;
    out.swap(m_pending);
    m_pending.clear();
    return true;
  }

private:
  std::mutex m_lock;
  std::vector<uint8_t> m_pending;
};

class Receiver
{
public:
  Receiver(const fmd_params& p, double tuner_freq, const char* adapter_name)
    : m_if_rate(p.sample_rate_if), m_tuner_hz(tuner_freq), m_adapter(adapter_name ? adapter_name : "")
  {
  }
  ~Receiver()
  {
    if (m_decoder)
      fmd_destroy(m_decoder);
  }
  Receiver(const Receiver&) = delete;
  Receiver& operator=(const Receiver&) = delete;

  /* decoder + stream state of a freshly opened live stream (RadioReceiver.cpp:296-349): a stream
   * change is announced first, the clock starts at one STREAM_TIME_BASE, the decoder is reset */
  int Open(const fmd_params& p)
  {
    fmd_callbacks cb{};
    cb.add_uecp_frame = [](void* self, unsigned, const uint8_t* f, unsigned n) -> int {
      return static_cast<Receiver*>(self)->m_rds.append_frame(f, n) ? 1 : 0;
    };
    cb.set_channel_name = [](void* self, unsigned, const char name[9]) -> int {
      static_cast<Receiver*>(self)->set_station_name(name);
      return 1; // no settings dialog in this host: the name is always accepted
    };
    cb.is_setting_active = [](void*, unsigned) -> int { return 0; };
    const int rc = fmd_create(&p, &cb, this, &m_decoder);
    if (rc != FMD_OK)
      return rc;
    m_announce_change.store(true);
    m_clock = double(FMD_STREAM_TIME_BASE);
    // page-locked blocks for the first deliveries (RTL-SDR byte blocks: 2 bytes per sample; sized by the
    // source's default block, RTL_SDR_Source.h:25)
    m_blocks.prime(4, size_t(FMD_MAX_BLOCK) * 2);
    return fmd_reset(m_decoder);
  }

  /* source side */
  bool Write(const void* data, unsigned samples, bool bytes_in)
  {
    if (!samples)
      return true; // empty deliveries are ignored (RadioReceiver.cpp:428)
    return m_blocks.push(data, size_t(samples) * (bytes_in ? 2 : 8), samples, bytes_in);
  }
  void End() { m_blocks.close(); }
  size_t QueuedSamples() { return m_blocks.backlog(); }
  void SetStreamChange() { m_announce_change.store(true); }

  /* consumer side: 1 = packet, 0 = source ended and everything was delivered, < 0 = decoder error */
  int NextPacket(fmd_demux_packet* pkt)
  {
    *pkt = fmd_demux_packet{};
    if (m_announce_change.exchange(false))
    {
      pkt->stream_id = FMD_STREAM_CHANGE;
      return 1;
    }
    if (m_rds.take_all(m_payload))
    { // everything the group decoder produced since the last read, stamped with the clock as it is
      pkt->stream_id = FMD_STREAM_RDS;
      pkt->data = m_payload.data();
      pkt->size = int(m_payload.size());
      pkt->pts = m_clock;
      return 1;
    }
    // "Input buffer is growing (system too slow)": more than 10 s of IQ waiting (:509-513)
    if (!m_backlog_warned && double(m_blocks.backlog()) > 10.0 * m_if_rate)
      m_backlog_warned = true;

    IqBlock* blk = m_blocks.pop();
    if (!blk)
      return 0;
    // the caller-visible buffer has the reference's size: 2 floats per IQ sample (:519-520)
    m_payload.resize(size_t(blk->samples) * 2 * sizeof(float));
    float* pcm = reinterpret_cast<float*>(m_payload.data());
    int floats;
    {
      std::lock_guard<std::mutex> g(m_status_lock); // status read-outs see whole calls only
      floats = blk->bytes_in
                   ? fmd_process_stream_u8(m_decoder, static_cast<const uint8_t*>(blk->mem), blk->samples, pcm)
                   : fmd_process_stream(m_decoder, static_cast<const float*>(blk->mem), blk->samples, pcm);
    }
    m_blocks.recycle(blk);
    if (floats < 0)
      return floats;
    const double span = double(floats) * FMD_STREAM_TIME_BASE / 2 / 48000;
    pkt->stream_id = FMD_STREAM_AUDIO;
    pkt->data = m_payload.data();
    pkt->size = int(size_t(floats) * sizeof(float));
    pkt->duration = span;
    pkt->pts = m_clock;
    m_clock += span;
    return 1;
  }

  /* the three values of GetSignalStatus(float&, float&, bool&); false while no stream is running
   * or a stream change is still to be announced */
  bool Levels(float& if_db, float& audio_db, bool& stereo)
  {
    Snapshot s;
    if (!snapshot(s))
      return false;
    if_db = s.if_db;
    audio_db = s.audio_db;
    stereo = s.st.stereo_detected != 0;
    return true;
  }

  /* PVRSignalStatus.  The reference's format string has five conversions for six arguments, so its
   * "IF=" shows the tuned frequency in MHz, "BB=" the interface level and "Audio=" the baseband
   * level; hosts parse that text, so it is reproduced as is. */
  bool PvrStatus(fmd_pvr_signal_status& out)
  {
    Snapshot s;
    if (!snapshot(s))
      return false;
    out = fmd_pvr_signal_status{};
    const double tuned_mhz = (m_tuner_hz + s.st.tuning_offset) * 1.0e-6;
    const double bb_db = 20 * std::log10(s.st.baseband_level) + 3.01;
    std::snprintf(out.adapter_status, sizeof(out.adapter_status),
                  "Freq.=%8.4fMHz - %s - IF=%+5.1fdB  BB=%+5.1fdB  Audio=%+5.1fdB", m_tuner_hz / 1000000,
                  s.st.stereo_detected ? "Stereo" : "Mono", tuned_mhz, s.if_db, bb_db);
    std::snprintf(out.adapter_name, sizeof(out.adapter_name), "%s", m_adapter.c_str());
    std::snprintf(out.provider_name, sizeof(out.provider_name), "%s", s.station.c_str());
    out.signal = int(2.5 * (s.if_db + 40) * 656);
    out.snr = int((s.audio_db + 100) * 656);
    return true;
  }

  fmd_decoder* Decoder() { return m_decoder; }
  bool BacklogWarned() const { return m_backlog_warned; }

private:
  struct Snapshot
  {
    fmd_status st{};
    float if_db = 0, audio_db = 0;
    std::string station;
  };
  bool snapshot(Snapshot& s)
  {
    std::lock_guard<std::mutex> g(m_status_lock);
    if (!m_decoder || m_announce_change.load())
      return false;
    if (fmd_get_status(m_decoder, &s.st) != FMD_OK)
      return false;
    fmd_audio_level lvl{};
    (void)fmd_batch_get_audio_level(fmd_decoder_batch(m_decoder), 0, &lvl);
    s.if_db = 20 * std::log10(s.st.interface_level);
    s.audio_db = 20 * std::log10(lvl.level) + 3.01;
    s.station = m_station;
    return true;
  }
  /* called from inside the decode call (the status lock is held by this thread's NextPacket) */
  void set_station_name(const char name[9])
  {
    std::string t(name);
    const char* ws = " \t\n\v\f\r";
    const size_t a = t.find_first_not_of(ws);
    const size_t z = t.find_last_not_of(ws);
    m_station = a == std::string::npos ? std::string() : t.substr(a, z - a + 1);
  }

  static fmd_batch* fmd_decoder_batch(fmd_decoder* d);

  const double m_if_rate;
  const double m_tuner_hz;
  const std::string m_adapter;

  fmd_decoder* m_decoder = nullptr;
  BlockPool m_blocks;
  UecpBytes m_rds;
  std::atomic<bool> m_announce_change{false};
  double m_clock = 0.0; // pts of the next packet
  bool m_backlog_warned = false;

  std::mutex m_status_lock; // decode call vs. status read-outs from another thread
  std::string m_station;    // written inside the decode call, read under m_status_lock

  std::vector<uint8_t> m_payload; // data of the packet handed out last
};

} // namespace fmd
