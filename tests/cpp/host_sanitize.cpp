// Host-only code of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are
// not available on the pool): the filter design for a sweep of constructor parameters (csrc/fmd_design.hpp: what
// cFmDecoder's constructor computes, FmDecode.cpp:237-314) and the UECP group decoder on arbitrary groups
// (csrc/fmd_groups.hpp: RDSGroupDecoder.cpp:166-1001).  Built and run by tests/test_host_sanitizers.py.
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "../../pvr.rtl.radiofm_amd/csrc/fmd_design.hpp"
#include "../../pvr.rtl.radiofm_amd/csrc/fmd_groups.hpp"

static unsigned long long g_frames = 0, g_bytes = 0, g_names = 0;
static int on_frame(void*, unsigned, const uint8_t* p, unsigned n)
{
  for (unsigned i = 0; i < n; i++)
    g_bytes += p[i]; // (every byte of the frame is read: a short buffer would show)
  g_frames++;
  return 1;
}
static int on_name(void*, unsigned, const char* name)
{
  for (int i = 0; i < 8; i++)
    g_bytes += (unsigned char)name[i];
  g_names++;
  return 1;
}
static int on_active(void*, unsigned) { return 0; }

int main(int argc, char** argv)
{
  const bool full = argc > 1 && argv[1][0] == 'f'; // "full": the whole sweep (90 s); else a tenth of it
  // ---- designs ----
  unsigned made = 0, refused = 0, seen = 0;
  const double rates[] = {250e3, 400e3, 1.0e6, 1.2e6, 1.8e6, 2.048e6, 2.4e6, 3.2e6, 6.4e6, 10e6, 12e6, 16.2e6};
  const double pcms[] = {38100.0, 40000.0, 44100.0, 48000.0, 96000.0, 192000.0, 32000.0};
  const double bws[] = {15000.0, 10000.0, 17000.0, 21000.0};
  const unsigned orders[] = {0, 1, 2, 7, 88, 257, 4096, 8192};
  const unsigned tables[] = {0, 1, 64, 256, 1000};
  for (double fs : rates)
    for (unsigned D = 1; D <= 56; D += (D < 16 ? 1 : 5))
      for (double pcm : pcms)
        for (double bw : bws)
          for (unsigned order : orders)
            for (unsigned table : tables)
            {
              if (!full && seen++ % 10 != 0)
                continue;
              if ((made + refused) % 7 != 0 && order > 88) // (keep the long-filter designs to a seventh of the sweep)
              {
                refused++;
                continue;
              }
              fmd::Params p;
              p.sample_rate_if = fs;
              p.tuning_offset = (int(made % 9) - 4) * 0.1 * fs;
              p.sample_rate_pcm = pcm;
              p.bandwidth_pcm = bw;
              p.downsample = D;
              p.us_version = made & 1;
              p.table_size = table;
              p.if_filter_order = order;
              try
              {
                const fmd::Design d = fmd::make_design(p);
                g_bytes += d.if_coeff.size() + d.lpf_taps.size() + d.rds_lpf_taps.size();
                made++;
              }
              catch (const std::exception&)
              {
                refused++;
              }
            }
  // ---- group decoder ----
  fmd_callbacks cb{};
  cb.add_uecp_frame = on_frame;
  cb.set_channel_name = on_name;
  cb.is_setting_active = on_active;
  fmd::GroupDecoder gd(&cb, nullptr, 0);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  uint16_t pi = 0x1234;
  const unsigned ngroups = full ? 2000000u : 200000u;
  for (unsigned k = 0; k < ngroups; k++)
  {
    const uint64_t r = rnd();
    if ((r & 0x3FF) == 0)
      pi = uint16_t(r >> 40);
    uint16_t b[4] = {pi, uint16_t(r >> 10), uint16_t(r >> 26), uint16_t(r >> 42)};
    if (k % 3 == 0) // text-like payloads with the control characters the text decoders look for
    {
      static const uint8_t ctl[8] = {0x0A, 0x0B, 0x0D, 0x1F, 0x00, 0xFF, 0xFE, 0xFD};
      auto ch = [&](unsigned x) { return uint8_t((x & 7) == 0 ? ctl[(x >> 3) & 7] : 0x20 + (x >> 3) % 0x5F); };
      b[2] = uint16_t(ch(unsigned(r >> 12)) << 8 | ch(unsigned(r >> 24)));
      b[3] = uint16_t(ch(unsigned(r >> 36)) << 8 | ch(unsigned(r >> 48)));
    }
    if (k % 5 == 0)
      b[1] = uint16_t((b[1] & 0x0FFF) | (((k / 5) % 16) << 12)); // every type in turn
    gd.push(b);
    if ((r & 0xFFFFF) == 1)
      gd.reset();
  }
  std::printf("designs made %u refused %u; groups %u -> frames %llu names %llu (checksum %llu)\n", made, refused,
              ngroups, g_frames, g_names, g_bytes);
  return made > 1000 && g_frames > ngroups ? 0 : 1;
}
