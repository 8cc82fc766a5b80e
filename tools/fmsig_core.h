/*
 * fmsig_core.h -- per-sample formulas of the synthetic FM generator, shared by the host
 * generator (tools/fmsig.c) and the HIP generator (csrc/fmsig_device.hip).
 * Test / bench infrastructure; see fmsig.h.
 */
#ifndef FMSIG_CORE_H
#define FMSIG_CORE_H
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define FMSIG_HD __host__ __device__ static inline
#else
#define FMSIG_HD static inline
#endif

#define FMSIG_2PI 6.283185307179586476925286766559
#define FMSIG_BITRATE 1187.5

typedef struct fmsig_chan
{
  double inv_fs;
  double f_offset, dev, amp;
  double a_mono, a_stereo, a_pilot, a_rds;
  double f_left, f_right;
  double noise_sigma;
  uint64_t seed;
} fmsig_chan;

FMSIG_HD double fmsig_frac(double x)
{
  return x - floor(x);
}
/* sin / cos of 2*pi*(f*t) with the cycle count reduced first */
FMSIG_HD double fmsig_sinc(double f, double t)
{
  return sin(FMSIG_2PI * fmsig_frac(f * t));
}
FMSIG_HD double fmsig_cosc(double f, double t)
{
  return cos(FMSIG_2PI * fmsig_frac(f * t));
}

FMSIG_HD uint64_t fmsig_mix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* integral of m(t) from 0 to t (seconds), dbits = differential RDS bit table */
FMSIG_HD double fmsig_int_m(const fmsig_chan* c, double t, uint64_t n, const uint8_t* dbits,
                            unsigned period_bits)
{
  const double w = 1.0 / FMSIG_2PI;
  double acc = 0.0;
  /* mono: a/2 * [ (1-cos wL t)/wL + (1-cos wR t)/wR ] */
  acc += 0.5 * c->a_mono *
         ((1.0 - fmsig_cosc(c->f_left, t)) * w / c->f_left +
          (1.0 - fmsig_cosc(c->f_right, t)) * w / c->f_right);
  /* stereo: a/2 * (sin wL t - sin wR t) * sin w38 t, product-to-sum, integrated */
  if (c->a_stereo != 0.0)
  {
    const double f38 = 38000.0;
    double l = 0.5 * (fmsig_sinc(f38 - c->f_left, t) * w / (f38 - c->f_left) -
                      fmsig_sinc(f38 + c->f_left, t) * w / (f38 + c->f_left));
    double r = 0.5 * (fmsig_sinc(f38 - c->f_right, t) * w / (f38 - c->f_right) -
                      fmsig_sinc(f38 + c->f_right, t) * w / (f38 + c->f_right));
    acc += 0.5 * c->a_stereo * (l - r);
  }
  if (c->a_pilot != 0.0)
    acc += c->a_pilot * (1.0 - fmsig_cosc(19000.0, t)) * w / 19000.0;
  if (c->a_rds != 0.0)
  {
    /* bit index from the integer sample index to avoid boundary rounding:
     * k = floor(n * 1187.5 / fs); every complete bit integrates to zero
     * (57 kHz = 48 x bit rate), so only the current bit contributes. */
    (void)n;
    double cyc = t * FMSIG_BITRATE;
    uint64_t k = (uint64_t)floor(cyc);
    double s = dbits[k % period_bits] ? 1.0 : -1.0;
    const double f1 = 47.0 * FMSIG_BITRATE, f2 = 49.0 * FMSIG_BITRATE;
    acc += c->a_rds * s * 0.5 * (fmsig_sinc(f1, t) * w / f1 - fmsig_sinc(f2, t) * w / f2);
  }
  return acc;
}

/* one IQ sample, quantised to two u8 */
FMSIG_HD void fmsig_sample_u8(const fmsig_chan* c, uint64_t n, const uint8_t* dbits,
                              unsigned period_bits, uint8_t* out_i, uint8_t* out_q)
{
  double t = (double)n * c->inv_fs;
  double cycles = fmsig_frac(c->f_offset * t) + c->dev * fmsig_int_m(c, t, n, dbits, period_bits);
  double ph = FMSIG_2PI * fmsig_frac(cycles);
  double vi = c->amp * cos(ph);
  double vq = c->amp * sin(ph);
  if (c->noise_sigma > 0.0)
  {
    uint64_t h = fmsig_mix64(c->seed ^ (n * 0xD6E8FEB86659FD93ull));
    uint64_t h2 = fmsig_mix64(h);
    double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740993.0); /* (0,1) */
    double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    double r = c->noise_sigma * sqrt(-2.0 * log(u1));
    vi += r * cos(FMSIG_2PI * u2);
    vq += r * sin(FMSIG_2PI * u2);
  }
  double qi = floor((vi + 1.0) * 127.5 + 0.5);
  double qq = floor((vq + 1.0) * 127.5 + 0.5);
  qi = qi < 0.0 ? 0.0 : (qi > 255.0 ? 255.0 : qi);
  qq = qq < 0.0 ? 0.0 : (qq > 255.0 ? 255.0 : qq);
  *out_i = (uint8_t)qi;
  *out_q = (uint8_t)qq;
}

/* RTL_SDR_Source.cpp:209-210: float(u8 / (255.0/2.0) - 1.0) */
FMSIG_HD float fmsig_u8_to_float(uint8_t v)
{
  return (float)((double)v / (255.0 / 2.0) - 1.0);
}

#endif
