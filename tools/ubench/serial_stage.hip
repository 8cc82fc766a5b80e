// The serial stage (k_demod_serial: FM PLL wave + pilot / RDS-oscillator wave) alone on the chip, on a
// synthetic FM stereo+RDS baseband: time per launch, shader cycles per workgroup, and a hash of
// everything the kernel wrote (rows of br / mix, final channel state).  Dev aid for work on the two
// sample loops: build it once per variant (-D switches of csrc/fmd_kernels.hip.h / fmd_math.h), run
// the binaries side by side -- equal hashes = the variant computes the same bits.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/serial_stage.hip -o serial_stage
//   ./serial_stage [channels=8192] [launches=6] [M=5958]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pvr.rtl.radiofm_amd/csrc/fmd_design.hpp"
#include "../../pvr.rtl.radiofm_amd/csrc/fmd_kernels.hip.h"

#define CK(x)                                                                        \
  do                                                                                 \
  {                                                                                  \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess)                                                            \
    {                                                                                \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; i++)
    h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

int main(int argc, char** argv)
{
  const unsigned C = argc > 1 ? unsigned(atoi(argv[1])) : 8192u;
  const int launches = argc > 2 ? atoi(argv[2]) : 6;
  const unsigned M = argc > 3 ? unsigned(atoi(argv[3])) : 5958u;
  const unsigned CP = (C + 63u) & ~63u;
  fmd::Params p{};
  p.sample_rate_if = 2.4e6;
  p.tuning_offset = -0.15 * 2.4e6;
  p.sample_rate_pcm = 48000.0;
  p.bandwidth_pcm = 15000.0;
  p.downsample = 11;
  const fmd::Design d = fmd::make_design(p);
  const unsigned Mstride = (M + 1 + 15u) & ~15u;

  // one station's IF-FIR output: unit-ish phasor, FM by (L+R) + pilot + (L-R) on 38 kHz + RDS-like
  // 57 kHz tone, slightly noisy; channel c gets it rotated and scaled a little (lanes differ)
  std::vector<float> base(size_t(2) * Mstride);
  {
    double ph = 0.3;
    uint32_t lcg = 12345u;
    const double fb = d.fs_bb;
    for (unsigned n = 0; n < Mstride; n++)
    {
      const double t = n / fb;
      const double m = 0.35 * 0.5 * (sin(2 * M_PI * 1000 * t) + sin(2 * M_PI * 2500 * t)) +
                       0.35 * 0.5 * (sin(2 * M_PI * 1000 * t) - sin(2 * M_PI * 2500 * t)) * sin(2 * M_PI * 38000 * t) +
                       0.09 * sin(2 * M_PI * 19000 * t) + 0.06 * sin(2 * M_PI * 57000 * t) * sin(2 * M_PI * 1187.5 * t);
      ph += 2 * M_PI * 75000.0 * m / fb;
      lcg = lcg * 1664525u + 1013904223u;
      const double n1 = ((lcg >> 8) & 0xffff) / 65536.0 - 0.5;
      lcg = lcg * 1664525u + 1013904223u;
      const double n2 = ((lcg >> 8) & 0xffff) / 65536.0 - 0.5;
      base[2 * n] = float(0.9 * cos(ph) + 0.01 * n1);
      base[2 * n + 1] = float(0.9 * sin(ph) + 0.01 * n2);
    }
  }
  std::vector<float> demod(size_t(2) * (size_t(Mstride) * C + fmd::DS));
  for (unsigned c = 0; c < C; c++)
  {
    const float g = 1.0f + 0.001f * float(c % 97), cr = cosf(0.01f * float(c % 61)), sr = sinf(0.01f * float(c % 61));
    for (unsigned n = 0; n < Mstride; n++)
    {
      const float re = base[2 * n], im = base[2 * n + 1];
      demod[2 * (size_t(c) * Mstride + n)] = g * (re * cr - im * sr);
      demod[2 * (size_t(c) * Mstride + n) + 1] = g * (re * sr + im * cr);
    }
  }

  float2 *d_demod, *d_br, *d_mix;
  float* d_f;
  int* d_i;
  uint16_t* d_r;
  double* d_tab;
  long long* d_probe;
  unsigned *h_err, *h_hs;
  const unsigned Hbb = d.rs_order, Hmix = unsigned(d.hb[0].len - 1);
  const size_t br_rows = size_t(2 * fmd::RS_B + Hbb + M + 1), mix_rows = size_t(Hmix + M + 1);
  CK(hipMalloc(&d_demod, demod.size() * 4));
  CK(hipMemcpy(d_demod, demod.data(), demod.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_br, br_rows * CP * 8));
  CK(hipMalloc(&d_mix, mix_rows * CP * 8));
  CK(hipMemset(d_br, 0, br_rows * CP * 8));
  CK(hipMemset(d_mix, 0, mix_rows * CP * 8));
  CK(hipMalloc(&d_f, size_t(fmd::F_SLOTS) * CP * 4));
  CK(hipMalloc(&d_i, size_t(fmd::I_SLOTS) * CP * 4));
  CK(hipMalloc(&d_r, size_t(4) * CP * 2));
  CK(hipMemset(d_f, 0, size_t(fmd::F_SLOTS) * CP * 4));
  CK(hipMemset(d_i, 0, size_t(fmd::I_SLOTS) * CP * 4));
  const std::vector<double>& tab = d.sincos_tab256;
  CK(hipMalloc(&d_tab, tab.size() * 8));
  CK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_probe, size_t(3) * (CP / 64) * 8));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h_err), 8, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h_hs), size_t(fmd::HS_WORDS) * CP * 4,
                   hipHostMallocMapped | hipHostMallocCoherent));
  h_err[0] = h_err[1] = 0;
  fmd::ChannelState st{};
  st.f = d_f;
  st.i = d_i;
  st.r_data = d_r;
  st.CP = CP;
  st.spin_limit = 1u << 20;
  void* dp = nullptr;
  CK(hipHostGetDevicePointer(&dp, h_err, 0));
  st.err = static_cast<unsigned*>(dp);
  CK(hipHostGetDevicePointer(&dp, h_hs, 0));
  st.hs = static_cast<unsigned*>(dp);
  {
    std::vector<float> v(CP, d.p_freq0);
    CK(hipMemcpy(st.F(fmd::F_P_FREQ), v.data(), CP * 4, hipMemcpyHostToDevice));
    std::fill(v.begin(), v.end(), 1.0f);
    CK(hipMemcpy(st.F(fmd::F_OSC_RE), v.data(), CP * 4, hipMemcpyHostToDevice));
  }
  fmd::DemodConsts k{};
  k.pll_alpha = d.pll_alpha;
  k.pll_beta = d.pll_beta;
  k.nco_hl = d.nco_hl;
  k.nco_ll = d.nco_ll;
  k.demod_gain = d.demod_gain;
  k.p_minfreq = d.p_minfreq;
  k.p_maxfreq = d.p_maxfreq;
  k.p_b0 = d.p_b0;
  k.p_a1 = d.p_a1;
  k.p_a2 = d.p_a2;
  k.p_lf_b0 = d.p_lf_b0;
  k.p_lf_b1 = d.p_lf_b1;
  k.p_minsignal = d.p_minsignal;
  k.p_lock_delay = d.p_lock_delay;
  k.osc_cos = d.rds_osc_cos;
  k.osc_sin = d.rds_osc_sin;
  const FmdSincosTab sct{d.sct_inv_h, d.sct_h_hi, d.sct_h_lo};
  const unsigned groups = CP / 64;
  float2* brp = d_br + size_t(2 * fmd::RS_B) * CP;

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<long long> probe(size_t(3) * groups);
  for (int form = 0; form < (getenv("SS_ALL_FORMS") ? 4 : 2); form++)
  { // 0: whole-CU form (two groups per workgroup), 1: shared form (one group per workgroup);
    // SS_ALL_FORMS=1 also runs 2: <2,false> and 3: <1,true> (what the register claim and the group count each cost)
    CK(hipMemset(d_f, 0, size_t(fmd::F_SLOTS) * CP * 4)); // both forms start from a fresh decoder
    CK(hipMemset(d_i, 0, size_t(fmd::I_SLOTS) * CP * 4));
    {
      std::vector<float> v(CP, d.p_freq0);
      CK(hipMemcpy(st.F(fmd::F_P_FREQ), v.data(), CP * 4, hipMemcpyHostToDevice));
      std::fill(v.begin(), v.end(), 1.0f);
      CK(hipMemcpy(st.F(fmd::F_OSC_RE), v.data(), CP * 4, hipMemcpyHostToDevice));
    }
    double ms_sum = 0, cyc_sum = 0, cyc_max = 0;
    int cnt = 0;
    for (int l = 0; l < launches; l++)
    {
      CK(hipMemset(d_probe, 0, probe.size() * 8));
      CK(hipEventRecord(e0, nullptr));
      if (form == 0)
        hipLaunchKernelGGL((fmd::k_demod_serial<2, true>), dim3((groups + 1) / 2), dim3(256), 0, nullptr, d_demod,
                           Mstride, M, C, CP, k, st, brp, Hbb, d_mix, Hmix, d_tab, sct, unsigned(l & 3), d_probe, 0.0f, 0.0f);
      else if (form == 1)
        hipLaunchKernelGGL((fmd::k_demod_serial<1, false>), dim3(groups), dim3(128), 0, nullptr, d_demod, Mstride,
                           M, C, CP, k, st, brp, Hbb, d_mix, Hmix, d_tab, sct, unsigned(l & 3), d_probe, 0.0f, 0.0f);
      else if (form == 2)
        hipLaunchKernelGGL((fmd::k_demod_serial<2, false>), dim3((groups + 1) / 2), dim3(256), 0, nullptr, d_demod,
                           Mstride, M, C, CP, k, st, brp, Hbb, d_mix, Hmix, d_tab, sct, unsigned(l & 3), d_probe, 0.0f, 0.0f);
      else
        hipLaunchKernelGGL((fmd::k_demod_serial<1, true>), dim3(groups), dim3(128), 0, nullptr, d_demod, Mstride,
                           M, C, CP, k, st, brp, Hbb, d_mix, Hmix, d_tab, sct, unsigned(l & 3), d_probe, 0.0f, 0.0f);
      CK(hipEventRecord(e1, nullptr));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(probe.data(), d_probe, probe.size() * 8, hipMemcpyDeviceToHost));
      if (l == 0)
        continue; // warm-up
      ms_sum += ms;
      cnt++;
      const unsigned nwg = (form == 0 || form == 2) ? (groups + 1) / 2 : groups;
      double s = 0, mx = 0;
      for (unsigned w = 0; w < nwg; w++)
      {
        const double cyc = double(probe[3 * w + 2] & 0xffffffffffll);
        s += cyc;
        mx = cyc > mx ? cyc : mx;
      }
      cyc_sum += s / nwg;
      cyc_max = mx > cyc_max ? mx : cyc_max;
    }
    // everything the kernel wrote
    std::vector<float> out_br(size_t(M) * CP * 2), out_mix(size_t(M) * CP * 2), sf(size_t(fmd::F_SLOTS) * CP);
    std::vector<int> si(size_t(fmd::I_SLOTS) * CP);
    CK(hipMemcpy(out_br.data(), brp + size_t(Hbb) * CP, out_br.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out_mix.data(), d_mix + size_t(Hmix) * CP, out_mix.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sf.data(), d_f, sf.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(si.data(), d_i, si.size() * 4, hipMemcpyDeviceToHost));
    uint64_t h = fnv(out_br.data(), out_br.size() * 4);
    h = fnv(out_mix.data(), out_mix.size() * 4, h);
    h = fnv(sf.data(), sf.size() * 4, h);
    h = fnv(si.data(), si.size() * 4, h);
    printf("%s  C=%u M=%u: %.4f ms per launch, %.4f Mcycles per workgroup (max %.4f), %.1f cycles/sample, "
           "err=%u, hash=%016llx\n",
           form == 0 ? "k_demod_serial<2,true> " : form == 1 ? "k_demod_serial<1,false>" : form == 2 ? "k_demod_serial<2,false>" : "k_demod_serial<1,true> ", C, M, ms_sum / cnt,
           cyc_sum / cnt / 1e6, cyc_max / 1e6, cyc_sum / cnt / M, h_err[0], (unsigned long long)h);
  }
  return 0;
}
