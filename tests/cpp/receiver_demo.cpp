/*
 * receiver_demo.cpp -- a caller written the way the reference's cRadioReceiver uses cFmDecoder
 * (src/RadioReceiver.cpp:296-300 construct, :349 Reset, :519-525 ProcessStream into a buffer of
 * samples*2 floats, :387-414 AddUECPDataFrame byte stuffing, :551-572 status getters), compiled
 * against include/fm_decoder.hpp and linked with libfmd_hip.so only (no torch, system ROCm).
 *
 * usage: receiver_demo <iq.f32> <fs> <downsample> <audio_out.f32> <uecp_out.bin>
 *   iq.f32 = interleaved float32 I/Q, processed in blocks of 65536 samples.
 */
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "fm_decoder.hpp"

class cRadioReceiver
{
public:
  bool AddUECPDataFrame(uint8_t* frame, unsigned int len)
  {
    if (m_uecp.size() > 16384) // RadioReceiver.cpp:389-390
      return false;
    std::vector<uint8_t> tmp(2 * len + 2);
    const int n = fmd_uecp_stuff_frame(frame, len, tmp.data(), unsigned(tmp.size()));
    m_uecp.insert(m_uecp.end(), tmp.begin(), tmp.begin() + n);
    m_frames++;
    return true;
  }
  bool SetChannelName(std::string name)
  {
    m_name = name;
    return true;
  }
  bool IsSettingActive() { return false; }

  std::vector<uint8_t> m_uecp;
  std::string m_name;
  unsigned m_frames = 0;
};

int main(int argc, char** argv)
{
  if (argc < 6)
  {
    std::fprintf(stderr, "usage: %s iq.f32 fs downsample audio_out.f32 uecp_out.bin\n", argv[0]);
    return 2;
  }
  const double if_rate = std::atof(argv[2]);
  const unsigned downsample = unsigned(std::atoi(argv[3]));
  FILE* fin = std::fopen(argv[1], "rb");
  FILE* faud = std::fopen(argv[4], "wb");
  if (!fin || !faud)
    return 2;

  cRadioReceiver receiver;
  cFmDecoder* decoder = nullptr;
  try
  {
    decoder = new cFmDecoder(&receiver, if_rate, -0.15 * if_rate, 48000.0, 15000.0, downsample);
  }
  catch (const std::exception& e)
  {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  decoder->Reset();

  const unsigned block = 65536; // cRtlSdrSource::default_block_length
  std::vector<ComplexType> iq(block);
  std::vector<float> audio(size_t(block) * 2); // AllocateDemuxPacket(iq.size() * sizeof(float) * 2)
  size_t total = 0;
  for (;;)
  {
    const size_t got = std::fread(iq.data(), sizeof(ComplexType), block, fin);
    if (got < 8192)
      break;
    const unsigned n = decoder->ProcessStream(iq.data(), unsigned(got), audio.data());
    std::fwrite(audio.data(), sizeof(float), n, faud);
    total += n;
  }
  std::fclose(fin);
  std::fclose(faud);
  if (FILE* fu = std::fopen(argv[5], "wb"))
  {
    std::fwrite(receiver.m_uecp.data(), 1, receiver.m_uecp.size(), fu);
    std::fclose(fu);
  }
  std::printf("floats=%zu stereo=%d pilot=%.6f if=%.6f bb=%.6f offset=%.3f frames=%u name=%s\n", total,
              int(decoder->StereoDetected()), decoder->GetPilotLevel(), decoder->GetInterfaceLevel(),
              decoder->GetBasebandLevel(), decoder->GetTuningOffset(), receiver.m_frames,
              receiver.m_name.c_str());
  delete decoder;
  return 0;
}
