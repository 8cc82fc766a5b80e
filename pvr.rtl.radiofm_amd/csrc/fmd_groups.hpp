/*
 * fmd_groups.hpp -- host-side RDS group -> UECP frame decoder (integer byte work; stays on
 * the CPU by design).  Behaviour follows cRDSGroupDecoder (RDSGroupDecoder.cpp:136-1001):
 * group types 0A/0B, 1A/1B, 2A/2B, 3A, 4A, 8A, 10A and the RT+ / TFC open-data applications
 * produce UECP message frames ADD(2) SQC MFL payload CRC16(2); the rest are accepted and
 * ignored like the reference's empty decoders.  Deviation on purpose: the PS scratch text is
 * per decoder, not the function-static of RDSGroupDecoder.cpp:311 (which would make batched
 * channels interfere).  Members the reference never initialises start at zero.
 */
#pragma once

#include <cstdint>
#include <cstring>

#include "../../include/fmd.h"

namespace fmd
{

inline uint16_t uecp_crc16(const uint8_t* p, int len) // CRC16-CCITT, RDSGroupDecoder.cpp:961-977
{
  uint16_t crc = 0xffff;
  while (len--)
  {
    crc = uint16_t((crc >> 8) | (crc << 8));
    crc ^= *p++;
    crc ^= uint16_t((crc & 0xff) >> 4);
    crc ^= uint16_t((crc << 8) << 4);
    crc ^= uint16_t(((crc & 0xff) << 4) << 1);
  }
  return uint16_t(~crc);
}

class GroupDecoder
{
public:
  GroupDecoder(const fmd_callbacks* cb, void* user, unsigned channel) : user_(user), channel_(channel)
  {
    if (cb)
      cb_ = *cb;
    reset();
  }

  void reset() // RDSGroupDecoder.cpp:136-164
  {
    pi_ = 0;
    rt_segreg_ = 0;
    rt_count_ = 0;
    rt_first_ = false;
    di_ = 0;
    di_prev_ = 0xff;
    ms_ = 0;
    ms_prev_ = 0xff;
    pin_ = 0xffff;
    ptyn_set_ = 0;
    ps_set_ = 0;
    ta_tp_ = -1;
    rtp_ready_ = false;
    std::memset(rt_, 0, sizeof(rt_));
    std::memset(oda_, 0, sizeof(oda_));
    std::memset(ptyn_, 0x20, sizeof(ptyn_));
    std::memset(ps_name_, 0x20, sizeof(ps_name_));
  }

  void push(const uint16_t b[4]) // DecodeRDS, RDSGroupDecoder.cpp:166-269
  {
    const unsigned gt = (b[1] >> 11) & 0x1F;
    const bool ver_b = gt & 1;
    if (b[0] != pi_)
    {
      reset();
      pi_ = b[0];
      begin(0x01);
      put(0);
      put(1);
      put(pi_ & 0xff);
      put(pi_ >> 8);
      send();
    }
    const int pty = (b[1] >> 5) & 0x1F;
    if (pty != pty_)
    {
      pty_ = pty;
      begin(0x07);
      put(0);
      put(1);
      put(uint8_t(pty));
      send();
    }
    switch (gt >> 1)
    {
      case 0:
        type0(b);
        break;
      case 1:
        type1(b, ver_b);
        break;
      case 2:
        type2(b, ver_b);
        break;
      case 3:
        if (!ver_b)
          type3a(b);
        else
          oda_or_nothing(b, gt);
        break;
      case 4:
        if (!ver_b)
          type4a(b);
        else
          oda_or_nothing(b, gt);
        break;
      case 8:
        if (!ver_b && oda_[gt] <= 0)
          type8a(b);
        else
          oda_or_nothing(b, gt);
        break;
      case 10:
        if (!ver_b)
          type10a(b);
        else
          oda_or_nothing(b, gt);
        break;
      case 14:
      case 15:
        break; // EON / RBDS / fast switching: empty in the reference (:864-901)
      default:
        oda_or_nothing(b, gt);
        break;
    }
  }

private:
  fmd_callbacks cb_{};
  void* user_;
  unsigned channel_;
  uint8_t seq_ = 0;
  int fill_ = 0;
  uint8_t frame_[263] = {};
  int oda_[32] = {};
  int pty_ = 0;
  int ta_tp_ = -1;
  char ptyn_[9] = {};
  int ptyn_ab_ = 0;
  int ptyn_set_ = 0;
  uint8_t di_ = 0, di_prev_ = 0xff;
  int di_count_ = 0;
  uint8_t ms_ = 0, ms_prev_ = 0xff;
  char ps_name_[9] = {};
  char ps_text_[9] = {};
  int ps_set_ = 0;
  uint16_t pin_ = 0xffff;
  uint16_t pi_ = 0;
  char rt_[66] = {};
  bool rt_first_ = false;
  int rt_ab_ = 0;
  uint32_t rt_segreg_ = 0;
  int rt_count_ = 0;
  bool rtp_ready_ = false;

  bool setting_active() const { return cb_.is_setting_active && cb_.is_setting_active(user_, channel_); }

  void begin(uint8_t mec) // ClearUECPFrame + first AddStuffingValue
  {
    frame_[0] = 0;
    frame_[1] = 0;
    frame_[2] = seq_;
    frame_[3] = 0;
    fill_ = 0;
    put(mec);
  }
  void put(unsigned v)
  {
    if (fill_ > 255)
      return;
    frame_[4 + fill_++] = uint8_t(v);
  }
  void send() // SendUECPFrame :979-991
  {
    if (setting_active())
      return;
    seq_++;
    frame_[3] = uint8_t(fill_);
    const uint16_t crc = uecp_crc16(frame_, fill_ + 4);
    frame_[4 + fill_] = uint8_t(crc >> 8);
    frame_[5 + fill_] = uint8_t(crc & 0xff);
    if (cb_.add_uecp_frame)
      cb_.add_uecp_frame(user_, channel_, frame_, unsigned(fill_ + 6));
  }

  void oda_or_nothing(const uint16_t* b, unsigned gt)
  {
    if (oda_[gt] == 0x4bd7)
    { // RT+ tags, only right after a completed radiotext (:910-928)
      if (!rtp_ready_)
        return;
      begin(0x46);
      put(8);
      put(0x4b);
      put(0xd7);
      for (int k = 1; k < 4; k++)
      {
        put(b[k] >> 8);
        put(b[k] & 0xff);
      }
      send();
      rtp_ready_ = false;
    }
    else if (oda_[gt] == 0xcd46)
    { // TFC (:929-941)
      begin(0x46);
      put(7);
      put(0xcd);
      put(0x46);
      put(b[1] & 0xff);
      put(b[2] >> 8);
      put(b[2] & 0xff);
      put(b[3] >> 8);
      put(b[3] & 0xff);
      send();
    }
  }

  void type0(const uint16_t* b) // :309-422
  {
    const unsigned seg = b[1] & 3;
    const uint8_t dibit = uint8_t(8 >> seg); // segment 0 carries d3 ... segment 3 carries d0
    if (b[1] & 4)
      di_ |= dibit;
    else
      di_ &= uint8_t(~dibit);
    di_count_++;
    const int tatp = ((b[1] & 0x10) ? 1 : 0) | ((b[1] & 0x400) ? 2 : 0);
    if (tatp != ta_tp_)
    {
      ta_tp_ = tatp;
      begin(0x03);
      put(0);
      put(1);
      put(unsigned(tatp));
      send();
    }
    if (di_count_ >= 4 && di_prev_ != di_)
    {
      di_count_ = 0;
      di_prev_ = di_;
      begin(0x04);
      put(0);
      put(1);
      put(di_ & 0xf);
      send();
    }
    ms_ = (b[1] & 8) ? 1 : 0;
    if (ms_prev_ != ms_)
    {
      ms_prev_ = ms_;
      begin(0x05);
      put(0);
      put(1);
      put(ms_);
      send();
    }
    ps_text_[2 * seg] = char(b[3] >> 8);
    ps_text_[2 * seg + 1] = char(b[3] & 0xff);
    ps_set_ |= 1 << seg;
    if (ps_set_ == 0xF && (setting_active() || std::memcmp(ps_name_, ps_text_, 8) != 0))
    {
      const bool accepted = cb_.set_channel_name ? cb_.set_channel_name(user_, channel_, ps_text_) != 0 : true;
      if (accepted)
      {
        begin(0x02);
        put(0);
        put(1);
        for (int i = 0; i < 8; i++)
          put(uint8_t(ps_text_[i]));
        send();
        std::memcpy(ps_name_, ps_text_, 8);
      }
      ps_set_ = 0;
    }
  }

  void type1(const uint16_t* b, bool ver_b) // :554-588
  {
    if (pin_ != b[3])
    {
      pin_ = b[3];
      begin(0x06);
      put(0);
      put(1);
      put(pin_ >> 8);
      put(pin_ & 0xff);
      send();
    }
    if (!ver_b)
    {
      begin(0x1A);
      put(0);
      put((b[2] >> 8) & 0x7F);
      put(b[2] & 0xff);
      send();
    }
  }

  void type2(const uint16_t* b, bool ver_b) // :593-659
  {
    const unsigned ptr = b[1] & 0xf;
    rtp_ready_ = false;
    if (ptr == 0 && rt_first_ && rt_count_ > 1)
    {
      bool ready = true;
      for (int i = 0; i < rt_count_; i++)
        if (!(rt_segreg_ & (1u << i)))
        {
          ready = false;
          rt_segreg_ = 0;
          rt_count_ = 0;
          break;
        }
      if (ready)
      {
        begin(0x0A);
        put(0);
        put(1);
        put(65);
        put(unsigned(rt_ab_));
        for (int i = 0; i < 64; i++)
          put(uint8_t(rt_[i]));
        send();
        rtp_ready_ = true;
      }
    }
    const int ab = (b[1] >> 4) & 1;
    if (rt_ab_ != ab)
    {
      std::memset(rt_, 0x20, sizeof(rt_));
      rt_ab_ = ab;
      rt_first_ = false;
      rt_segreg_ = 0;
      rt_count_ = 0;
    }
    if (!ver_b)
    {
      rt_[ptr * 4] = char(b[2] >> 8);
      rt_[ptr * 4 + 1] = char(b[2] & 0xff);
      rt_[ptr * 4 + 2] = char(b[3] >> 8);
      rt_[ptr * 4 + 3] = char(b[3] & 0xff);
    }
    else
    {
      rt_[ptr * 2] = char(b[3] >> 8);
      rt_[ptr * 2 + 1] = char(b[3] & 0xff);
    }
    rt_segreg_ |= 1u << ptr;
    rt_count_++;
    if (!rt_first_ && ptr == 0)
      rt_first_ = true;
  }

  void type3a(const uint16_t* b) // :664-704
  {
    begin(0x40);
    put(b[1] & 0x1F);
    put(b[3] >> 8);
    put(b[3] & 0xff);
    put(0);
    put(b[2] >> 8);
    put(b[2] & 0xff);
    put(0);
    send();
    const unsigned target = b[1] & 0x1F;
    oda_[target] = (b[3] == 0x4bd7 || b[3] == 0xcd46) ? int(b[3]) : 0;
  }

  void type4a(const uint16_t* b) // :709-737
  {
    const double mjd = double(((b[1] & 0x03) << 15) | ((b[2] >> 1) & 0x7fff));
    const unsigned hours = ((b[2] & 1u) << 4) | ((b[3] >> 12) & 0xf);
    const unsigned minutes = (b[3] >> 6) & 0x3f;
    const int offset = b[3] & 0x3f;
    unsigned year = unsigned(int((mjd - 15078.2) / 365.25));
    unsigned month = unsigned(int((mjd - 14956.1 - int(year * 365.25)) / 30.6001));
    const unsigned day = unsigned(mjd - 14956 - int(year * 365.25) - int(month * 30.6001));
    const int K = (month == 14 || month == 15) ? 1 : 0;
    year += unsigned(K + 1900);
    month -= unsigned(1 + K * 12);
    begin(0x0D);
    put(year % 100);
    put(month);
    put(day);
    put(hours);
    put(minutes);
    put(0);
    put(0);
    put(unsigned(offset));
    send();
  }

  void type8a(const uint16_t* b) // :780-794
  {
    begin(0x30);
    put(6);
    put(0);
    put(b[1] & 0x1F);
    put(b[2] >> 8);
    put(b[2] & 0xff);
    put(b[3] >> 8);
    put(b[3] & 0xff);
    send();
  }

  void type10a(const uint16_t* b) // :812-844
  {
    const unsigned ptr = b[1] & 1;
    const int ab = (b[1] >> 4) & 1;
    if (ptyn_ab_ != ab)
    {
      std::memset(ptyn_, 0x20, 8);
      ptyn_ab_ = ab;
      ptyn_set_ = 0;
    }
    ptyn_[ptr * 4] = char(b[2] >> 8);
    ptyn_[ptr * 4 + 1] = char(b[2] & 0xff);
    ptyn_[ptr * 4 + 2] = char(b[3] >> 8);
    ptyn_[ptr * 4 + 3] = char(b[3] & 0xff);
    ptyn_set_ |= 1 << ptr;
    if (ptyn_set_ & 3)
    {
      begin(0x3A);
      put(0);
      put(1);
      for (int i = 0; i < 8; i++)
        put(uint8_t(ptyn_[i]));
      send();
    }
  }
};

} // namespace fmd
