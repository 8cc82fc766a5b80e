/*
 * dropin_decl_order.cpp -- compile-only check (g++ -fsyntax-only, no GPU, no link) that
 * include/fm_decoder.hpp drops into the reference's declaration order:
 *   RadioReceiver.h:23    class cFmDecoder;                 (forward declaration, seen first)
 *   RadioReceiver.h:124   cFmDecoder* m_FMDecoder = nullptr; (member of cRadioReceiver)
 *   RadioReceiver.cpp:8-11  #include "RadioReceiver.h" ... #include "FmDecode.h"  <- replaced
 *   RadioReceiver.cpp:296-300  m_FMDecoder = new cFmDecoder(this, if_rate, offset, pcm, bw, downsample);
 *   RadioReceiver.cpp:349  m_FMDecoder->Reset();
 *   RadioReceiver.cpp:373  delete m_FMDecoder;
 *   RadioReceiver.cpp:524-525, :551-553, :565-572  ProcessStream and the getters
 * The class below has the reference's member names but none of its code.
 */
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

// ---- what RadioReceiver.h declares before any decoder header is seen ----
class cFmDecoder;

class cRadioReceiver
{
public:
  bool OpenLiveStream();
  void CloseLiveStream();
  bool DemuxOnce(const std::vector<std::complex<float>>& iqsamples, float* pData);
  bool GetSignalStatus(float& interfaceLevel, bool& stereo);
  bool AddUECPDataFrame(uint8_t* UECPDataFrame, unsigned int length);
  bool SetChannelName(std::string name);
  bool IsSettingActive();

private:
  cFmDecoder* m_FMDecoder = nullptr;
  double m_IfRate = 2.4e6, m_PCMRate = 48000.0;
  double m_activeChannelFrequency = 100.0e6, m_activeTunerFreq = 100.36e6;
};

// ---- RadioReceiver.cpp:11, the one changed line ----
#include "fm_decoder.hpp"

bool cRadioReceiver::OpenLiveStream()
{
  double bandwidth_pcm = DEFAULT_BANDWIDTH_PCM;
  unsigned int downsample = 11;
  m_FMDecoder = new cFmDecoder(this, m_IfRate, m_activeChannelFrequency - m_activeTunerFreq, m_PCMRate,
                               bandwidth_pcm, downsample);
  m_FMDecoder->Reset();
  return true;
}

void cRadioReceiver::CloseLiveStream()
{
  if (m_FMDecoder)
  {
    delete m_FMDecoder;
    m_FMDecoder = nullptr;
  }
}

bool cRadioReceiver::DemuxOnce(const std::vector<std::complex<float>>& iqsamples, float* pData)
{
  unsigned int iSize = m_FMDecoder->ProcessStream(iqsamples.data(), iqsamples.size(), pData);
  return iSize != 0;
}

bool cRadioReceiver::GetSignalStatus(float& interfaceLevel, bool& stereo)
{
  interfaceLevel = m_FMDecoder->GetInterfaceLevel();
  stereo = m_FMDecoder->StereoDetected();
  return m_FMDecoder->GetTuningOffset() + m_FMDecoder->GetBasebandLevel() + m_FMDecoder->GetPilotLevel() >= 0;
}

bool cRadioReceiver::AddUECPDataFrame(uint8_t*, unsigned int) { return true; }
bool cRadioReceiver::SetChannelName(std::string) { return true; }
bool cRadioReceiver::IsSettingActive() { return false; }
