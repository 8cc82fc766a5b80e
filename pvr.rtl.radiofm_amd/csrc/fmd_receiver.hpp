/*
 * fmd_receiver.hpp -- the stream side of cRadioReceiver around the decoder (host code).
 *
 * Mirrors, member for member, what the reference does between its RTL-SDR source thread and
 * Kodi's demuxer (all file:line relative to /root/reference/src/):
 *   OpenLiveStream's stream state        RadioReceiver.cpp:296-349
 *   AddUECPDataFrame (byte stuffing)     RadioReceiver.cpp:387-414
 *   SourceQueuedSamples                  RadioReceiver.cpp:420-424
 *   WriteDataBuffer / EndDataBuffer      RadioReceiver.cpp:426-443
 *   SourceGetSamples                     RadioReceiver.cpp:445-460
 *   DemuxRead                            RadioReceiver.cpp:462-542
 *   GetSignalStatus (both)               RadioReceiver.cpp:544-582
 *   SetChannelName                       RadioReceiver.cpp:600-612 (no settings dialog here)
 * The decoder is an fmd_decoder (GPU); the audio level meter of DemuxRead (:526-528) runs on the
 * device with the audio (k_audio_tail) and is read back through fmd_batch_get_audio_level.
 */
#pragma once

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fmd.h"

namespace fmd
{

class Receiver
{
public:
  struct Block
  {
    std::vector<uint8_t> bytes; // complex<float> or (I,Q) byte pairs
    unsigned samples = 0;
    bool u8 = false;
  };

  Receiver(const fmd_params& p, double tuner_freq, const char* adapter_name)
    : m_IfRate(p.sample_rate_if), m_activeTunerFreq(tuner_freq),
      m_adapterName(adapter_name ? adapter_name : "")
  {
  }
  ~Receiver()
  {
    if (m_FMDecoder)
      fmd_destroy(m_FMDecoder);
  }

  /* RadioReceiver.cpp:296-300, :345-349 */
  int Open(const fmd_params& p)
  {
    fmd_callbacks cb{};
    cb.add_uecp_frame = &Receiver::OnFrame;
    cb.set_channel_name = &Receiver::OnName;
    cb.is_setting_active = &Receiver::OnActive;
    int rc = fmd_create(&p, &cb, this, &m_FMDecoder);
    if (rc != FMD_OK)
      return rc;
    m_StreamChange = true;
    m_PTSNext = FMD_STREAM_TIME_BASE;
    return fmd_reset(m_FMDecoder);
  }

  /* :387-414 */
  bool AddUECPDataFrame(const uint8_t* frame, unsigned length)
  {
    if (m_UECPOutputBuffer.size() > 16384)
      return false;
    std::unique_lock<std::mutex> lock(m_UECPMutex);
    m_UECPOutputBuffer.push_back(0xFE);
    for (unsigned i = 0; i < length; i++)
    {
      const uint8_t value = frame[i];
      if (value < 0xFD)
        m_UECPOutputBuffer.push_back(value);
      else
      {
        m_UECPOutputBuffer.push_back(0xFD);
        m_UECPOutputBuffer.push_back(uint8_t((value & 3) - 1));
      }
    }
    m_UECPOutputBuffer.push_back(0xFF);
    return true;
  }

  /* :600-612 with m_SettingsDialog == nullptr; StringUtils::Trim = isspace on both ends */
  bool SetChannelName(std::string name)
  {
    std::unique_lock<std::mutex> lock(m_AudioSignalMutex);
    auto notspace = [](char c) { return !::isspace((unsigned char)c); };
    name.erase(name.begin(), std::find_if(name.begin(), name.end(), notspace));
    name.erase(std::find_if(name.rbegin(), name.rend(), notspace).base(), name.end());
    m_channelName = name;
    return true;
  }

  /* :420-424 */
  size_t SourceQueuedSamples()
  {
    std::unique_lock<std::mutex> lock(m_AudioSourceMutex);
    return m_AudioSourceSize;
  }

  /* :426-436 */
  void WriteDataBuffer(Block&& blk)
  {
    if (blk.samples)
    {
      std::unique_lock<std::mutex> lock(m_AudioSourceMutex);
      m_AudioSourceSize += blk.samples;
      m_AudioSourceBuffer.push_back(std::move(blk));
      if (m_AudioSourceBuffer.size() > 3)
        m_AudioSourceEvent.notify_one();
    }
  }

  /* :438-443 */
  void EndDataBuffer()
  {
    std::unique_lock<std::mutex> lock(m_AudioSourceMutex);
    m_AudioSourceEndMarked = true;
    m_AudioSourceEvent.notify_all();
  }

  void SetStreamChange() { m_StreamChange = true; }

  /* :462-542.  Returns 1 with *pkt filled, 0 for the reference's nullptr, < 0 on a decoder error. */
  int DemuxRead(fmd_demux_packet* pkt)
  {
    std::memset(pkt, 0, sizeof(*pkt));
    if (m_StreamChange)
    { // :471-477
      pkt->stream_id = FMD_STREAM_CHANGE;
      m_StreamChange = false;
      return 1;
    }
    {
      std::unique_lock<std::mutex> lock(m_UECPMutex);
      if (!m_UECPOutputBuffer.empty())
      { // :482-503
        m_packet.assign(m_UECPOutputBuffer.begin(), m_UECPOutputBuffer.end());
        pkt->data = m_packet.data();
        pkt->stream_id = FMD_STREAM_RDS;
        pkt->size = int(m_packet.size());
        pkt->pts = m_PTSNext;
        m_UECPOutputBuffer.clear();
        return 1;
      }
    }
    // :510-514 "Input buffer is growing (system too slow)"
    if (!m_AudioSourceBufferWarning && SourceQueuedSamples() > 10 * m_IfRate)
      m_AudioSourceBufferWarning = true;

    Block blk;
    if (!SourceGetSamples(blk))
      return 0;
    m_packet.resize(size_t(blk.samples) * sizeof(float) * 2); // :519-520
    float* audio = reinterpret_cast<float*>(m_packet.data());
    const int iSize =
        blk.u8 ? fmd_process_stream_u8(m_FMDecoder, blk.bytes.data(), blk.samples, audio)
               : fmd_process_stream(m_FMDecoder, reinterpret_cast<const float*>(blk.bytes.data()),
                                    blk.samples, audio);
    if (iSize < 0)
      return iSize;
    // :526-528: SamplesMeanRMS over the packet and the level average were computed with the audio
    const double duration = (double)(iSize)*FMD_STREAM_TIME_BASE / 2 / 48000; // :531
    pkt->data = m_packet.data();
    pkt->stream_id = FMD_STREAM_AUDIO;
    pkt->size = int(iSize * sizeof(float));
    pkt->duration = duration;
    pkt->pts = m_PTSNext;
    m_PTSNext = m_PTSNext + duration;
    return 1;
  }

  /* :544-556 */
  bool GetSignalStatus(float& interfaceLevel, float& audioLevel, bool& stereo)
  {
    std::unique_lock<std::mutex> lock(m_AudioSignalMutex);
    if (!m_FMDecoder || m_StreamChange)
      return false;
    fmd_status st{};
    if (fmd_get_status(m_FMDecoder, &st) != FMD_OK)
      return false;
    interfaceLevel = 20 * std::log10(st.interface_level);
    audioLevel = 20 * std::log10(AudioLevel()) + 3.01;
    stereo = st.stereo_detected != 0;
    return true;
  }

  /* :558-582.  The format string has five conversions for six arguments, so "IF=" shows the tuned
   * frequency, "BB=" the interface level and "Audio=" the baseband level; kept as is. */
  bool GetSignalStatus(fmd_pvr_signal_status& out)
  {
    std::unique_lock<std::mutex> lock(m_AudioSignalMutex);
    if (!m_FMDecoder || m_StreamChange)
      return false;
    fmd_status st{};
    if (fmd_get_status(m_FMDecoder, &st) != FMD_OK)
      return false;
    const float interfaceLevel = 20 * std::log10(st.interface_level);
    const float audioLevel = 20 * std::log10(AudioLevel()) + 3.01;
    std::memset(&out, 0, sizeof(out));
    std::snprintf(out.adapter_status, sizeof(out.adapter_status),
                  "Freq.=%8.4fMHz - %s - IF=%+5.1fdB  BB=%+5.1fdB  Audio=%+5.1fdB",
                  m_activeTunerFreq / 1000000, st.stereo_detected ? "Stereo" : "Mono",
                  (m_activeTunerFreq + st.tuning_offset) * 1.0e-6, interfaceLevel,
                  20 * std::log10(st.baseband_level) + 3.01);
    std::snprintf(out.adapter_name, sizeof(out.adapter_name), "%s", m_adapterName.c_str());
    std::snprintf(out.provider_name, sizeof(out.provider_name), "%s", m_channelName.c_str());
    out.signal = int(2.5 * (interfaceLevel + 40) * 656);
    out.snr = int((audioLevel + 100) * 656);
    return true;
  }

  fmd_decoder* Decoder() { return m_FMDecoder; }
  bool BufferWarning() const { return m_AudioSourceBufferWarning; }

private:
  /* :445-460 */
  bool SourceGetSamples(Block& samples)
  {
    std::unique_lock<std::mutex> lock(m_AudioSourceMutex);
    while (m_AudioSourceBuffer.empty() && !m_AudioSourceEndMarked)
      m_AudioSourceEvent.wait_for(lock, std::chrono::milliseconds(20));
    if (!m_AudioSourceBuffer.empty())
    {
      m_AudioSourceSize -= m_AudioSourceBuffer.front().samples;
      std::swap(samples, m_AudioSourceBuffer.front());
      m_AudioSourceBuffer.pop_front();
      return true;
    }
    return false;
  }

  float AudioLevel() // m_AudioLevel
  {
    fmd_audio_level a{};
    (void)fmd_batch_get_audio_level(fmd_decoder_batch(m_FMDecoder), 0, &a);
    return a.level;
  }

  static int OnFrame(void* user, unsigned, const uint8_t* frame, unsigned len)
  {
    return static_cast<Receiver*>(user)->AddUECPDataFrame(frame, len) ? 1 : 0;
  }
  static int OnName(void* user, unsigned, const char name[9])
  {
    return static_cast<Receiver*>(user)->SetChannelName(std::string(name)) ? 1 : 0;
  }
  static int OnActive(void*, unsigned) { return 0; } // IsSettingActive: no dialog

  static fmd_batch* fmd_decoder_batch(fmd_decoder* d);

  double m_IfRate;
  double m_activeTunerFreq;
  std::string m_adapterName;
  std::string m_channelName;
  bool m_StreamChange = false;
  fmd_decoder* m_FMDecoder = nullptr;

  std::mutex m_UECPMutex;
  std::vector<uint8_t> m_UECPOutputBuffer;
  std::mutex m_AudioSignalMutex;
  double m_PTSNext = 0.0;

  size_t m_AudioSourceSize = 0;
  std::deque<Block> m_AudioSourceBuffer;
  std::mutex m_AudioSourceMutex;
  std::condition_variable m_AudioSourceEvent;
  bool m_AudioSourceEndMarked = false;
  bool m_AudioSourceBufferWarning = false;

  std::vector<uint8_t> m_packet; // pData of the packet handed out last
};

} // namespace fmd
