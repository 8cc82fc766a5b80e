/*
 * fmd_design.hpp -- host-side constants and filter taps of the MI355X FM decoder.
 *
 * Everything cFmDecoder's constructor chain computes once per tune
 * (FmDecode.cpp:237-314 and the constructors it reaches) is evaluated here on the host
 * with the reference's float/double promotion pattern, then uploaded to the GPU.
 * Compile with -ffp-contract=off.  Citations are relative to /root/reference/src/.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace fmd
{

constexpr double K_2PI = 2.0 * 3.14159265358979323846; // Definitions.h:60
constexpr double K_PI = 3.14159265358979323846;

struct Params
{
  double sample_rate_if = 0;
  double tuning_offset = 0;
  double sample_rate_pcm = 48000.0;
  double bandwidth_pcm = 15000.0;
  unsigned downsample = 1;
  bool us_version = false;
  unsigned table_size = 0;      // 0 -> 64 (FmDecode.cpp:249)
  unsigned if_filter_order = 0; // 0 -> 8*downsample (FmDecode.cpp:262)
  int fir_reduction = 0;        // 0 sequential tap order (bit-exact), 1 shuffle-reduced, 2 fused multiply-add (opt-in)
};

struct Biquad
{
  float b0, b1, b2, a1, a2;
};

struct HalfBand
{
  int len;                 // taps; len - 1 = the rows of delay line in front of a stage's input
  std::vector<float> coef;
  bool cic = false;        // CCicN3DecimateBy2 (DownConvert.cpp:690-727) instead of a half-band: len = 3, i.e. two rows
                           // of delay line (m_Xeven, m_Xodd), no coefficient table
};

struct Design
{
  // cFmDecoder scalars
  float fs_if, fs_bb;
  unsigned D;
  unsigned table_size;
  float freq_dev;
  float demod_gain;
  float pll_alpha, pll_beta, nco_hl, nco_ll;
  float de_alpha;
  // cDownsampleFilter (IF)
  unsigned if_order;
  std::vector<float> if_coeff; // order+2 entries, [0] and [order+1] are 0
  // cDownsampleFilter (mono / stereo, identical design)
  unsigned rs_order;
  std::vector<float> rs_coeff;
  float rs_step;
  // cPilotPhaseLock
  float p_minfreq, p_maxfreq, p_b0, p_a1, p_a2, p_lf_b0, p_lf_b1, p_freq0, p_minsignal;
  int p_lock_delay;
  // audio tail
  std::vector<float> lpf_taps;
  Biquad notch;
  // RDS
  float rds_rate;
  float rds_nco_inc, rds_osc_cos, rds_osc_sin;
  std::vector<HalfBand> hb;
  std::vector<float> rds_lpf_taps;
  std::vector<float> rds_mf_taps;
  float rds_pll_alpha, rds_pll_beta, rds_nco_hl, rds_nco_ll;
  Biquad bitsync;
  // (sin, cos)(k * 2pi / 1024) for the NCO evaluation (fmd_math.h: fmd_sincos_tab)
  std::vector<double> sincos_tab;
  double sct_inv_h, sct_h_hi, sct_h_lo;
  // (sin, cos)(k / 256), k = 0 .. 2047 (fmd_math.h: fmd_sincos_p256)
  std::vector<double> sincos_tab256;
};

// cFineTuner table (FmDecode.cpp:45-58): 2*(cos, sin) of ((shift*i) % size) * step
inline std::vector<float> make_tuner_lut(unsigned table_size, int freq_shift)
{
  std::vector<float> lut(2 * table_size);
  const float phase_step = float(K_2PI / double(float(table_size)));
  for (unsigned i = 0; i < table_size; ++i)
  {
    const int64_t r = (int64_t(freq_shift) * int64_t(i)) % int64_t(table_size);
    const float phi = float(r) * phase_step;
    lut[2 * i] = cosf(phi) * 2.0f;
    lut[2 * i + 1] = sinf(phi) * 2.0f;
  }
  return lut;
}

inline int tuning_shift_for(const Params& p)
{
  const unsigned ts = p.table_size ? p.table_size : 64;
  return int(lrint(-double(ts) * p.tuning_offset / p.sample_rate_if)); // FmDecode.cpp:250
}

// MakeLanczosCoeff through the cDownsampleFilter ctor (DownConvert.cpp:18-56, :78)
inline std::vector<float> make_lanczos(unsigned order, double cutoff)
{
  const unsigned fo = order - 1;
  std::vector<float> c(fo + 3, 0.0f);
  double ysum = 0.0;
  for (int i = 1; i <= int(fo) + 1; i++)
  {
    const int t2 = int(2u * unsigned(i) - fo);
    double y = 1.0;
    if (t2 != 0)
    {
      const double x1 = cutoff * t2;
      const double x2 = t2 / double(fo + 2);
      y = (double(sinf(float(K_PI * x1))) / K_PI / x1) * (double(sinf(float(K_PI * x2))) / K_PI / x2);
    }
    c[i] = float(y);
    ysum += y;
  }
  for (unsigned i = 1; i <= fo + 1; i++)
    c[i] = float(double(c[i]) / ysum);
  return c;
}

// cFirFilter::Izero (FirFilter.cpp:39-58)
inline float bessel_i0(float x)
{
  const float x2 = x / 2.0f;
  float sum = 1.0f, ds = 1.0f, di = 1.0f;
  const float errorlimit = float(1e-9);
  do
  {
    float t = x2 / di;
    t *= t;
    ds *= t;
    sum += ds;
    di = float(double(di) + 1.0);
  } while (ds >= errorlimit * sum);
  return sum;
}

// cFirFilter::InitLPFilter with NumTaps = 0 (FirFilter.cpp:78-148)
inline std::vector<float> make_kaiser_lp(float scale, float astop, float fpass, float fstop, float fs)
{
  const float nfpass = fpass / fs;
  const float nfstop = fstop / fs;
  const float nfcut = (nfstop + nfpass) / 2.0f;
  float beta;
  if (astop < 20.96f)
    beta = 0;
  else if (astop >= 50.0f)
    beta = float(.1102 * double(astop - 8.71f));
  else
    beta = float(.5842 * double(powf(astop - 20.96f, float(0.4))) + double(.07886f * (astop - 20.96f)));
  unsigned ntaps = unsigned(double(astop - 8.0f) / (double(2.285f) * K_2PI * double(nfstop - nfpass)) + 1);
  if (ntaps > 75)
    ntaps = 75; // MAX_NUMCOEF, FirFilter.h:15
  if (ntaps < 3)
    ntaps = 3;
  const float fcenter = float(.5 * double(float(ntaps - 1)));
  const float izb = bessel_i0(beta);
  std::vector<float> taps(ntaps);
  for (unsigned n = 0; n < ntaps; ++n)
  {
    float x = float(n) - fcenter;
    float c;
    if (float(n) == fcenter)
      c = float(2.0 * double(nfcut));
    else
      c = float(double(sinf(float(K_2PI * double(x) * double(nfcut)))) / (K_PI * double(x)));
    x = (float(n) - (float(ntaps) - 1.0f) / 2.0f) / ((float(ntaps) - 1.0f) / 2.0f);
    taps[n] = scale * c * bessel_i0(beta * sqrtf(1 - (x * x))) / izb;
  }
  return taps;
}

enum BiquadType
{
  BQ_LP,
  BQ_HP,
  BQ_BP,
  BQ_BR
};

// cIirFilter::Init (IirFilter.cpp:11-60)
inline Biquad make_biquad(BiquadType type, float f0, float q, float fs)
{
  const float w0 = float(K_2PI * double(f0) / double(fs));
  const float alpha = float(double(sinf(w0)) / (2.0 * double(q)));
  const float A = float(1.0 / (1.0 + double(alpha)));
  const double cw = double(cosf(w0));
  Biquad b{};
  b.a1 = float(double(A) * (-2.0 * cw));
  b.a2 = float(double(A) * (1.0 - double(alpha)));
  switch (type)
  {
    case BQ_LP:
      b.b0 = float(double(A) * ((1.0 - cw) / 2.0));
      b.b1 = float(double(A) * (1.0 - cw));
      b.b2 = b.b0;
      break;
    case BQ_HP:
      b.b0 = float(double(A) * ((1.0 + cw) / 2.0));
      b.b1 = float(double(-A) * (1.0 + cw));
      b.b2 = b.b0;
      break;
    case BQ_BP:
      b.b0 = A * alpha;
      b.b1 = 0.0f;
      b.b2 = A * -alpha;
      break;
    case BQ_BR:
      b.b0 = float(double(A) * 1.0);
      b.b1 = float(double(A) * (-2.0 * cw));
      b.b2 = b.b0;
      break;
  }
  return b;
}

namespace detail
{
// distinct side taps h[0], h[2], ... of the CuteSDR half-band prototypes
// (filtercoef.h:62-150, Moe Wheatley, Simplified BSD); centre tap is 0.5
struct HbProto
{
  int len;
  double max_bw; // filtercoef.h:45-56
  std::vector<double> side;
};

inline const std::vector<HbProto>& hb_protos()
{
  static const std::vector<HbProto> t = {
      {11, .5 - .475, {0.0060431029837374152, -0.049372515458761493, 0.29332944952052842}},
      {15, .5 - .451, {-0.001442203300285281, 0.013017512802724852, -0.061653278604903369, 0.30007792316024057}},
      {19, .5 - .428, {0.00042366527106480427, -0.0040717333369021894, 0.019895653881950692, -0.070740034412329067, 0.30449249772844139}},
      {23, .5 - .409, {-0.00014987651418332164, 0.0014748633283609852, -0.0074416944990005314, 0.026163522731980929, -0.077593699116544707, 0.30754683719791986}},
      {27, .5 - .392, {0.000063730426952664685, -0.00061985193978569082, 0.0031512504783365756, -0.011173151342856621, 0.03171888754393197, -0.082917863582770729, 0.3097770473566307}},
      {31, .5 - .378, {-0.000030957335326552226, 0.00029271992847303054, -0.0014770381124258423, 0.0052539088990950535, -0.014856378748476874, 0.036406651919555999, -0.08699862567952929, 0.31140967076042625}},
      {35, .5 - .366, {0.000017017718072971716, -0.00015425042851962818, 0.00076219685751140838, -0.002691614694785393, 0.0075927497927344764, -0.018325727896057686, 0.040351004914363969, -0.090198224668969554, 0.31264689763504327}},
      {39, .5 - .356, {-0.000010175082832074367, 0.000088036416015024345, -0.00042370835558387595, 0.0014772557414459019, -0.0041468438954260153, 0.0099579126901608011, -0.021433527104289002, 0.043598963493432855, -0.092695953625928404, 0.31358799113382152}},
      {43, .5 - .347, {0.0000067666739082756387, -0.000055275221547958285, 0.00025654074579418561, -0.0008748125689163153, 0.0024249876017061502, -0.0057775190656021748, 0.012299834239523121, -0.024244050662087069, 0.046354303503099069, -0.094729903598633314, 0.31433918020123208}},
      {47, .5 - .340, {-0.0000045298314172004251, 0.000035333704512843228, -0.00015934776420643447, 0.0005340788063118928, -0.0014667949695500761, 0.0034792089350833247, -0.0073794356720317733, 0.014393786384683398, -0.026586603160193314, 0.048538673667907428, -0.09629115286535718, 0.31490673428547367}},
      {51, .5 - .333, {0.0000033359253688981639, -0.000024584155158361803, 0.00010677777483317733, -0.00034890723143173914, 0.00094239127078189603, -0.0022118302078923137, 0.0046575030752162277, -0.0090130973415220566, 0.016383673864361164, -0.028697281101743237, 0.05043292242400841, -0.097611898315791965, 0.31538104435015801}},
  };
  return t;
}
} // namespace detail

inline Design make_design(const Params& p)
{
  if (!(p.sample_rate_if > 0) || !(p.sample_rate_pcm > 0))
    throw std::invalid_argument("fmd: sample rates must be positive");
  // The 19 kHz notch (cIirFilter::Init(ftBR, 19000, 5, pcm rate), FmDecode.cpp:285 / IirFilter.cpp:11-60) has its
  // centre at or beyond Nyquist there: alpha = sin(w0) / 2Q <= 0, the poles leave the unit circle, and the
  // reference's own audio runs away to NaN within a few blocks (cRadioReceiver only ever asks for 48 kHz,
  // RadioReceiver.cpp:185).  Garbage cannot be met bit for bit (the sign of an invalid-operation NaN is the
  // machine's): refused.
  if (!(p.sample_rate_pcm > 2.0 * 19000.0))
    throw std::invalid_argument("fmd: sample_rate_pcm must be above 38 kHz (the reference's 19 kHz notch filter is "
                                "unstable at and below it: its audio diverges to NaN)");
  Design d{};
  d.D = p.downsample ? p.downsample : 1;
  d.fs_if = float(p.sample_rate_if);
  d.fs_bb = float(p.sample_rate_if / d.D); // FmDecode.cpp:248
  d.table_size = p.table_size ? p.table_size : 64;
  d.freq_dev = float(60000.0); // FmDecode.h:23
  d.demod_gain = float(1.0 / (60000.0 / double(d.fs_bb) * K_2PI)); // :254

  d.if_order = p.if_filter_order ? p.if_filter_order : 8 * d.D;
  d.if_coeff = make_lanczos(d.if_order, 0.6 / d.D); // :262
  d.rs_order = unsigned(int(double(d.fs_bb) / 1000.0)); // :264
  d.rs_coeff = make_lanczos(d.rs_order, p.bandwidth_pcm / double(d.fs_bb));
  d.rs_step = float(double(d.fs_bb) / p.sample_rate_pcm); // DownConvert.cpp:204
  if (d.rs_order < 2 || d.if_order < 2)
    throw std::invalid_argument("fmd: filter order too small");

  { // cPilotPhaseLock ctor (FmDecode.cpp:88-140) with the arguments of :257-260
    const float freq = float(19000.0 / double(d.fs_bb));
    const float bw = 50 / d.fs_bb;
    d.p_minfreq = float(double(freq - bw) * K_2PI);
    d.p_maxfreq = float(double(freq + bw) * K_2PI);
    d.p_minsignal = 0.04f;
    d.p_lock_delay = int(20.0f / bw);
    const float p1 = float(std::exp(double(-1.146f * bw) * K_2PI));
    const float p2 = float(std::exp(double(-5.331f * bw) * K_2PI));
    d.p_a1 = -p1 - p2;
    d.p_a2 = p1 * p2;
    d.p_b0 = 1 + d.p_a1 + d.p_a2;
    d.p_lf_b0 = float(double(0.62f * bw) * K_2PI);
    d.p_lf_b1 = float(double(-d.p_lf_b0) * std::exp(-0.1153 * double(bw) * K_2PI));
    d.p_freq0 = float(double(freq) * K_2PI);
  }

  d.notch = make_biquad(BQ_BR, float(19000.0), 5, float(p.sample_rate_pcm)); // :285
  d.lpf_taps = make_kaiser_lp(1.0f, 60.0f, 15000.0f, float(1.4 * 15000.0), float(p.sample_rate_pcm)); // :286
  { // InitDeemphasis :340-346
    const float tc = p.us_version ? float(75E-6) : float(50E-6);
    const float sr = float(p.sample_rate_pcm);
    d.de_alpha = (1.0f - expf(-1.0f / (sr * tc)));
  }
  { // FM PLL constants :305-312
    const float fac = float(K_2PI / double(d.fs_bb));
    const float bandwidth = 0.85f * d.fs_bb;
    const float maxdev = 0.95f * (0.5f * d.fs_bb);
    d.nco_ll = (-maxdev) * fac;
    d.nco_hl = (+maxdev) * fac;
    d.pll_alpha = 0.125f * bandwidth * fac;
    d.pll_beta = (d.pll_alpha * d.pll_alpha) / 2.0f;
  }

  { // CRDSDownConvert::SetDataRate(fs_bb, 8000) (DownConvert.cpp:327-371)
    const auto& protos = detail::hb_protos();
    const float max_bw = 8000.0f;
    float f = d.fs_bb;
    while ((double(f) > (double(max_bw) / protos.back().max_bw)) && (double(f) > 7900.0 * 2.0))
    {
      if (d.hb.size() >= 9) // m_pDecimatorPtrs[MAX_DECSTAGES = 10], the last one stays null (DownConvert.h:63, 168)
        throw std::invalid_argument("fmd: baseband rate needs more decimate-by-2 stages than the reference's list holds");
      if (double(f) >= (double(max_bw) / (.5 - .4985)))
      { // CIC order 3 (:340-341): baseband rates from 5.33 MHz up (cRadioReceiver never asks: RadioReceiver.cpp:285)
        HalfBand h;
        h.len = 3;
        h.coef.assign(3, 0.0f);
        h.cic = true;
        d.hb.push_back(h);
        f = float(double(f) / 2.0);
        continue;
      }
      for (const auto& pr : protos)
      {
        if (double(f) >= (double(max_bw) / pr.max_bw))
        {
          HalfBand h;
          h.len = pr.len;
          h.coef.assign(size_t(pr.len), 0.0f);
          for (size_t k = 0; k < pr.side.size(); k++)
          {
            h.coef[2 * k] = float(pr.side[k]);
            h.coef[size_t(pr.len) - 1 - 2 * k] = float(pr.side[k]);
          }
          h.coef[size_t(pr.len - 1) / 2] = float(0.5);
          d.hb.push_back(h);
          break;
        }
      }
      f = float(double(f) / 2.0);
    }
    d.rds_rate = f;
    // SetFrequency(-57000) (DownConvert.cpp:311-320)
    const float nco_freq = float(-57000.0);
    d.rds_nco_inc = float(K_2PI * double(nco_freq) / double(d.fs_bb));
    d.rds_osc_cos = cosf(d.rds_nco_inc);
    d.rds_osc_sin = sinf(d.rds_nco_inc);
  }
  { // cRDSRxSignalProcessor ctor + Reset (RDSProcess.cpp:43-118)
    const double bitrate = 57000.0 / 48.0;
    const float norm = float(K_2PI / double(d.rds_rate));
    d.rds_nco_ll = float((double(0.0f) - 12.0) * double(norm));
    d.rds_nco_hl = float((double(0.0f) + 12.0) * double(norm));
    d.rds_pll_alpha = float(2.0 * 0.707 * 1.00 * double(norm));
    d.rds_pll_beta = float(double(d.rds_pll_alpha * d.rds_pll_alpha) / (4.0 * 0.707 * 0.707));
    const unsigned L = unsigned(double(d.rds_rate) / bitrate);
    std::vector<float> mc(2 * L + 1, 0.0f);
    for (int i = 0; i <= int(L); i++)
    {
      const float t = float(i) / d.rds_rate;
      const float x = float(double(t) * bitrate);
      const float x64 = float(64.0 * double(x));
      const double shape = (1.0 / (1.0 / double(x) - double(x64))) - (1.0 / (9.0 / double(x) - double(x64)));
      const double c = double(cosf(float(2.0 * K_2PI * double(x))));
      mc[unsigned(i) + L] = float(.75 * c * shape);
      mc[L - unsigned(i)] = float(-.75 * c * shape);
    }
    if (L == 0)
      throw std::invalid_argument("fmd: RDS matched filter length out of range");
    // first 2L of 2L+1 (:77); cFirFilter::InitConstFir keeps at most MAX_NUMCOEF = 75 of them (FirFilter.cpp:305-306:
    // RDS rates above 44.5 kHz, e.g. a 12 MS/s baseband: 78 -> 75)
    d.rds_mf_taps.assign(mc.begin(), mc.begin() + std::min<size_t>(2 * L, 75));
    d.rds_lpf_taps = make_kaiser_lp(1.0f, 40.0f, 2400.0f, float(1.3 * 2400.0), d.rds_rate);
    d.bitsync = make_biquad(BQ_BP, float(bitrate), 500, d.rds_rate);
  }
  { // NCO sin/cos table, built in long double so every entry is the correctly rounded double
    const long double twopi = 6.283185307179586476925286766559005768L;
    const int n = 1024;
    d.sincos_tab.resize(size_t(2) * n);
    for (int k = 0; k < n; k++)
    {
      const long double a = twopi * k / (long double)n;
      d.sincos_tab[size_t(2) * k] = double(sinl(a));
      d.sincos_tab[size_t(2) * k + 1] = double(cosl(a));
    }
    const long double h = twopi / (long double)n;
    d.sct_inv_h = double(1.0L / h);
    double hd = double(h);
    uint64_t u;
    std::memcpy(&u, &hd, 8);
    u &= ~((uint64_t(1) << 15) - 1); // 38 significant bits: k * h_hi is exact for |k| < 2^15
    std::memcpy(&hd, &u, 8);
    d.sct_h_hi = hd;
    d.sct_h_lo = double(h - (long double)hd);
    d.sincos_tab256.resize(size_t(2) * 2048);
    for (int k = 0; k < 2048; k++)
    {
      const long double a = (long double)k / 256.0L;
      d.sincos_tab256[size_t(2) * k] = double(sinl(a));
      d.sincos_tab256[size_t(2) * k + 1] = double(cosl(a));
    }
  }
  return d;
}

} // namespace fmd
