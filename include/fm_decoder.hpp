/*
 * fm_decoder.hpp -- the reference's cFmDecoder class surface on top of the C ABI (fmd.h).
 *
 * A caller written against src/FmDecode.h:91-225 of AlwinEsch/pvr.rtl.radiofm (in practice
 * cRadioReceiver, src/RadioReceiver.cpp:296-300, :349, :373, :524-525, :551-553, :565-572)
 * compiles against this header unchanged: same class name, constructor signature, method names,
 * argument meaning and return values.  The three upward calls the RDS group decoder makes into
 * cRadioReceiver (AddUECPDataFrame / SetChannelName / IsSettingActive, RadioReceiver.h:77,80,115)
 * are forwarded through fmd_callbacks to the `proc` object handed to the constructor.
 *
 * Header-only; link with libfmd_hip.so.  Errors of the GPU library (the reference has no error
 * path) are reported by throwing std::runtime_error from the constructor and by returning 0
 * audio samples from ProcessStream.
 */
#pragma once

#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "fmd.h"

typedef std::complex<float> ComplexType; // Definitions.h:44
typedef float RealType;                  // Definitions.h:45

#ifndef DEFAULT_BANDWIDTH_PCM
#define DEFAULT_BANDWIDTH_PCM 15000.0 // FmDecode.h:24
#endif

/* Receiver: any class with
 *   bool AddUECPDataFrame(uint8_t* frame, unsigned int len);
 *   bool SetChannelName(std::string name);
 *   bool IsSettingActive();
 * i.e. cRadioReceiver's members.  The class cFmDecoder at the bottom binds it to the class named
 * cRadioReceiver, like FmDecode.h does (:27). */
template <class Receiver>
class cFmDecoderT
{
public:
  cFmDecoderT(Receiver* proc,
              double sample_rate_if,
              double tuning_offset,
              double sample_rate_pcm,
              double bandwidth_pcm = DEFAULT_BANDWIDTH_PCM,
              unsigned int downsample = 1,
              bool USver = false)
    : m_proc(proc)
  {
    fmd_params p{};
    p.sample_rate_if = sample_rate_if;
    p.tuning_offset = tuning_offset;
    p.sample_rate_pcm = sample_rate_pcm;
    p.bandwidth_pcm = bandwidth_pcm;
    p.downsample = downsample;
    p.us_version = USver ? 1 : 0;
    fmd_callbacks cb{};
    cb.add_uecp_frame = &cFmDecoderT::OnFrame;
    cb.set_channel_name = &cFmDecoderT::OnName;
    cb.is_setting_active = &cFmDecoderT::OnActive;
    if (fmd_create(&p, &cb, this, &m_dec) != FMD_OK)
      throw std::runtime_error(std::string("cFmDecoder: ") + fmd_last_error());
  }

  virtual ~cFmDecoderT() { fmd_destroy(m_dec); }

  cFmDecoderT(const cFmDecoderT&) = delete;
  cFmDecoderT& operator=(const cFmDecoderT&) = delete;

  void Reset() { fmd_reset(m_dec); }

  /* FmDecode.h:135 -- returns the number of floats written to `audio` (2 per frame) */
  unsigned int ProcessStream(const ComplexType* samples_in, unsigned int samples, float* audio)
  {
    const int n = fmd_process_stream(m_dec, reinterpret_cast<const float*>(samples_in), samples, audio);
    return n > 0 ? static_cast<unsigned int>(n) : 0u;
  }

  /* Not in the reference: cRtlSdrSource::ReadAsyncCB's conversion (RTL_SDR_Source.cpp:206-211)
   * and ProcessStream in one call -- buf = 2 * samples bytes as librtlsdr delivers them. */
  unsigned int ProcessStreamU8(const uint8_t* buf, unsigned int samples, float* audio)
  {
    const int n = fmd_process_stream_u8(m_dec, buf, samples, audio);
    return n > 0 ? static_cast<unsigned int>(n) : 0u;
  }

  bool StereoDetected() const { return Status().stereo_detected != 0; }
  RealType GetTuningOffset() const { return Status().tuning_offset; }
  RealType GetInterfaceLevel() const { return Status().interface_level; }
  RealType GetBasebandLevel() const { return Status().baseband_level; }
  RealType GetPilotLevel() const { return Status().pilot_level; }

private:
  fmd_status Status() const
  {
    fmd_status st{};
    fmd_get_status(m_dec, &st);
    return st;
  }
  static int OnFrame(void* user, unsigned, const uint8_t* frame, unsigned len)
  {
    auto* self = static_cast<cFmDecoderT*>(user);
    return self->m_proc && self->m_proc->AddUECPDataFrame(const_cast<uint8_t*>(frame), len) ? 1 : 0;
  }
  static int OnName(void* user, unsigned, const char name[9])
  {
    auto* self = static_cast<cFmDecoderT*>(user);
    return self->m_proc && self->m_proc->SetChannelName(std::string(name, 8)) ? 1 : 0;
  }
  static int OnActive(void* user, unsigned)
  {
    auto* self = static_cast<cFmDecoderT*>(user);
    return self->m_proc && self->m_proc->IsSettingActive() ? 1 : 0;
  }

  Receiver* const m_proc;
  fmd_decoder* m_dec = nullptr;
};

#ifndef FMD_NO_CFMDECODER_ALIAS
/* The reference's name, as a real class: RadioReceiver.h:23 forward-declares `class cFmDecoder;`
 * and keeps a `cFmDecoder*` member (:124) before any decoder header is seen, so the name must be
 * a class (a typedef would conflict with that declaration).  cRadioReceiver may still be
 * incomplete here (FmDecode.h:27 only forward-declares it too): the base template's members that
 * call into it are instantiated where they are used -- `new cFmDecoder(this, ...)` at
 * RadioReceiver.cpp:296-300, where the class is complete. */
class cRadioReceiver; // FmDecode.h:27
class cFmDecoder : public cFmDecoderT<cRadioReceiver>
{
public:
  using cFmDecoderT<cRadioReceiver>::cFmDecoderT;
};
#endif
