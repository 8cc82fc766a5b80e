/*
 * fmd_oracle.c -- CPU restatement (plain C) of cFmDecoder::ProcessStream and the
 * classes below it.  TEST INFRASTRUCTURE ONLY -- see fmd_oracle.h for the rules
 * and for the "parity unpinned / known-answer pinned" statement.
 *
 * Conventions that matter for bit-faithfulness:
 *  - RealType is float, K_2PI/K_PI are double literals (Definitions.h:17,45,60-64):
 *    every expression below keeps the reference's float/double promotion pattern.
 *  - MSIN/MCOS/MEXP/MPOW/MSQRT are the float libm calls even on double arguments
 *    (Definitions.h:47-57).
 *  - Unqualified abs/atan2/sqrt/exp/fmod resolve under libstdc++ to
 *    fabsf/atan2f/sqrtf/exp(double)/fmod(double) for the argument types used.
 *  - PLL NCOs use the x87 fsincos instruction on x86 (FmDecode.cpp:167,386,
 *    RDSProcess.cpp:245).
 *  - Build with -ffp-contract=off: the reference is compiled for baseline x86-64
 *    (no FMA), sums are accumulated sequentially in the written order.
 *
 * All file:line citations are relative to /root/reference/src/.
 */
#include "fmd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define K_2PI (2.0 * 3.14159265358979323846) /* Definitions.h:60 */
#define K_PI (3.14159265358979323846)        /* Definitions.h:61 */
#define K_PI2 (K_PI / 2.0)                   /* Definitions.h:63 */

typedef struct
{
  float re, im;
} cf32;

/* ------------------------------------------------------------------------- */
/* x87 sin/cos of a float phase, rounded to float (FmDecode.cpp:167,386)      */
/* ------------------------------------------------------------------------- */
static inline void sincos_nco(float phase, float* s, float* c)
{
#if defined(__x86_64__) || defined(__i386__)
  float sv, cv;
  __asm__ volatile("fsincos" : "=t"(cv), "=u"(sv) : "0"(phase));
  *s = sv;
  *c = cv;
#else
  *s = sinf(phase); /* FmDecode.cpp:171-172 */
  *c = cosf(phase);
#endif
}

void fmo_sincos_x87(float phase, float* s, float* c)
{
  sincos_nco(phase, s, c);
}

float fmo_atan2f(float y, float x)
{
  return atan2f(y, x);
}

/* ------------------------------------------------------------------------- */
/* cFineTuner  (FmDecode.cpp:45-82)                                           */
/* ------------------------------------------------------------------------- */
typedef struct
{
  unsigned index;
  unsigned size;
  cf32* table;
} fine_tuner;

static void fine_tuner_init(fine_tuner* ft, unsigned table_size, int freq_shift)
{
  ft->index = 0;
  ft->size = table_size;
  ft->table = (cf32*)calloc(table_size, sizeof(cf32));
  /* :50  float phase_step = K_2PI / float(table_size); */
  float phase_step = (float)(K_2PI / (double)(float)table_size);
  for (unsigned i = 0; i < table_size; ++i)
  {
    /* :53  signed 64-bit remainder, then int64 -> float, float multiply */
    int64_t r = ((int64_t)freq_shift * (int64_t)i) % (int64_t)table_size;
    float phi = (float)r * phase_step;
    float pc = cosf(phi);
    float ps = sinf(phi);
    ft->table[i].re = pc * 2.0f; /* :56 amplitude x2 */
    ft->table[i].im = ps * 2.0f;
  }
}

static void fine_tuner_process(fine_tuner* ft, const cf32* in, cf32* out, unsigned n)
{
  unsigned k = ft->index;
  for (unsigned i = 0; i < n; ++i)
  {
    cf32 a = in[i], b = ft->table[k];
    /* std::complex<float> product: (ac - bd) + i(ad + bc), each op rounded */
    out[i].re = a.re * b.re - a.im * b.im;
    out[i].im = a.re * b.im + a.im * b.re;
    if (++k == ft->size)
      k = 0;
  }
  ft->index = k;
}

/* ------------------------------------------------------------------------- */
/* MakeLanczosCoeff + cDownsampleFilter  (DownConvert.cpp:18-256)             */
/* ------------------------------------------------------------------------- */
static float* lanczos_coeff(unsigned fo, double cutoff)
{
  /* fo is the argument MakeLanczosCoeff receives (= filter order - 1, :78).
   * Array has fo+3 zero-initialised entries (:20); entries 1..fo+1 are set. */
  float* c = (float*)calloc(fo + 3, sizeof(float));
  double ysum = 0.0;
  for (int i = 1; i <= (int)fo + 1; i++)
  {
    int t2 = (int)(2u * (unsigned)i - fo); /* :33 unsigned wrap-around then int */
    double y;
    if (t2 == 0)
      y = 1.0;
    else
    {
      double x1 = cutoff * t2;
      double x2 = t2 / (double)(fo + 2);
      /* :44 sinf on a double argument, the rest in double */
      y = ((double)sinf((float)(K_PI * x1)) / K_PI / x1) *
          ((double)sinf((float)(K_PI * x2)) / K_PI / x2);
    }
    c[i] = (float)y;
    ysum += y;
  }
  for (unsigned i = 1; i <= fo + 1; i++)
    c[i] = (float)((double)c[i] / ysum); /* :53 float /= double */
  return c;
}

unsigned fmo_design_lanczos(unsigned filter_order_arg, double cutoff, float* out, unsigned cap)
{
  /* filter_order_arg is the cDownsampleFilter order; returns order+2 entries. */
  float* c = lanczos_coeff(filter_order_arg - 1, cutoff);
  unsigned n = filter_order_arg + 2;
  for (unsigned i = 0; i < n && i < cap; i++)
    out[i] = c[i];
  free(c);
  return n;
}

typedef struct
{
  double downsample;
  unsigned downsample_int;
  unsigned pos_int;
  float pos_frac;
  float* coeff;
  unsigned order;
  float* st_real;
  cf32* st_cplx;
} dsf;

static void dsf_init(dsf* f, unsigned order, double cutoff, double downsample, int integer_factor)
{
  f->downsample = downsample;
  f->downsample_int = integer_factor ? (unsigned)lrint(downsample) : 0; /* :68 */
  f->pos_int = 0;
  f->pos_frac = 0;
  f->order = order;
  f->coeff = lanczos_coeff(order - 1, cutoff); /* :78 */
  f->st_cplx = (cf32*)calloc(order, sizeof(cf32));
  f->st_real = (float*)calloc(order, sizeof(float));
}

static void dsf_free(dsf* f)
{
  free(f->coeff);
  free(f->st_cplx);
  free(f->st_real);
}

/* complex, integer decimation (DownConvert.cpp:98-154) */
static unsigned dsf_process_cplx(dsf* f, const cf32* in, cf32* out, unsigned n)
{
  const unsigned order = f->order;
  const unsigned step = f->downsample_int;
  unsigned p = f->pos_int;
  unsigned produced = 0;
  for (; p < n; p += step, produced++)
  {
    float yr = 0.0f, yi = 0.0f;
    for (unsigned j = 1; j <= order; j++)
    {
      /* sample x[p-j]: from this block when j <= p, else from the saved tail
       * of the previous block at index order + p - j (:116-118, :127) */
      cf32 x = (j <= p) ? in[p - j] : f->st_cplx[order + p - j];
      float k = f->coeff[j];
      yr += x.re * k;
      yi += x.im * k;
    }
    out[produced].re = yr;
    out[produced].im = yi;
  }
  f->pos_int = p - n; /* :132 */

  if (n < order)
  { /* :136-145 shift then append */
    unsigned j = 0;
    for (unsigned i = n; i < order; i++)
      f->st_cplx[j++] = f->st_cplx[i];
    for (unsigned i = 0; i < n; i++)
      f->st_cplx[j++] = in[i];
  }
  else
  {
    for (unsigned i = 0; i < order; i++)
      f->st_cplx[i] = in[n - order + i];
  }
  return produced;
}

/* real, fractional step via linear interpolation of the tap table
 * (DownConvert.cpp:195-233) or integer step (:164-194) */
static unsigned dsf_process_real(dsf* f, const float* in, float* out, unsigned n)
{
  const unsigned order = f->order;
  unsigned produced = 0;
  if (f->downsample_int != 0)
  {
    unsigned p = f->pos_int;
    for (; p < n; p += f->downsample_int, produced++)
    {
      float y = 0.0f;
      for (unsigned j = 1; j <= order; j++)
      {
        float x = (j <= p) ? in[p - j] : f->st_real[order + p - j];
        y += x * f->coeff[j];
      }
      out[produced] = y;
    }
    f->pos_int = p - n;
  }
  else
  {
    float p = f->pos_frac;
    float pstep = (float)f->downsample; /* :204 double -> float */
    float pf = p;
    unsigned pi = (unsigned)(int)pf;
    while (pi < n)
    {
      float k1 = pf - (float)pi;
      float k0 = 1.0f - k1;
      float y = 0.0f;
      for (unsigned j = 0; j <= order; j++)
      {
        float k = f->coeff[j] * k0 + f->coeff[j + 1] * k1;
        float s = (j <= pi) ? in[pi - j] : f->st_real[order + pi - j];
        y += k * s;
      }
      out[produced] = y;
      produced++;
      pf = p + (float)produced * pstep; /* :224 */
      pi = (unsigned)(int)pf;
    }
    f->pos_frac = pf - (float)n; /* :230 */
    if (f->pos_frac < 0)
      f->pos_frac = 0;
  }

  if (n < order)
  {
    unsigned j = 0;
    for (unsigned i = n; i < order; i++)
      f->st_real[j++] = f->st_real[i];
    for (unsigned i = 0; i < n; i++)
      f->st_real[j++] = in[i];
  }
  else
  {
    for (unsigned i = 0; i < order; i++)
      f->st_real[i] = in[n - order + i];
  }
  return produced;
}

/* ------------------------------------------------------------------------- */
/* cPilotPhaseLock  (FmDecode.cpp:88-229)                                     */
/* ------------------------------------------------------------------------- */
typedef struct
{
  float minfreq, maxfreq;
  float b0, a1, a2;
  float i1, i2, q1, q2;
  float lf_b0, lf_b1, lf_x1;
  float freq, phase;
  float minsignal;
  float pilot_level;
  int lock_delay, lock_cnt;
} pilot_pll;

static void pilot_init(pilot_pll* p, float freq, float bandwidth, float minsignal)
{
  p->minfreq = (float)((double)(freq - bandwidth) * K_2PI); /* :104 */
  p->maxfreq = (float)((double)(freq + bandwidth) * K_2PI);
  p->minsignal = minsignal;
  p->lock_delay = (int)(20.0f / bandwidth); /* :109 */
  p->lock_cnt = 0;
  /* :114-115 float*float, then *double, double exp, narrowed on store */
  float p1 = (float)exp((double)(-1.146f * bandwidth) * K_2PI);
  float p2 = (float)exp((double)(-5.331f * bandwidth) * K_2PI);
  p->a1 = -p1 - p2;
  p->a2 = p1 * p2;
  p->b0 = 1 + p->a1 + p->a2;
  p->lf_b0 = (float)((double)(0.62f * bandwidth) * K_2PI);                  /* :121 */
  p->lf_b1 = (float)((double)(-p->lf_b0) * exp(-0.1153 * (double)bandwidth * K_2PI)); /* :122 */
  p->freq = (float)((double)freq * K_2PI);                                  /* :131 */
  p->phase = 0;
  p->i1 = p->i2 = p->q1 = p->q2 = 0;
  p->lf_x1 = 0;
  p->pilot_level = 0;
}

static int pilot_process(pilot_pll* p, const float* in, float* out, unsigned n)
{
  p->pilot_level = 1000.0f;
  for (unsigned i = 0; i < n; ++i)
  {
    float ps, pc;
    sincos_nco(p->phase, &ps, &pc);
    out[i] = 2 * ps * pc; /* :177 */
    float x = in[i];
    float pi_ = ps * x;
    float pq = pc * x;
    pi_ = p->b0 * pi_ - p->a1 * p->i1 - p->a2 * p->i2; /* :185 */
    pq = p->b0 * pq - p->a1 * p->q1 - p->a2 * p->q2;
    p->i2 = p->i1;
    p->i1 = pi_;
    p->q2 = p->q1;
    p->q1 = pq;
    float err;
    if (pi_ > fabsf(pq)) /* :194 float abs under libstdc++ */
      err = pq / pi_;
    else if (pq > 0)
      err = 1;
    else
      err = -1;
    p->pilot_level = (pi_ < p->pilot_level) ? pi_ : p->pilot_level; /* std::min :204 */
    p->freq += p->lf_b0 * err + p->lf_b1 * p->lf_x1;
    p->lf_x1 = err;
    {
      /* :211 std::max(minfreq, std::min(maxfreq, freq)) */
      float t = (p->freq < p->maxfreq) ? p->freq : p->maxfreq;
      p->freq = (p->minfreq < t) ? t : p->minfreq;
    }
    p->phase += p->freq;
    if ((double)p->phase > K_2PI) /* :215-216 compare and subtract in double */
      p->phase = (float)((double)p->phase - K_2PI);
  }
  if (2 * p->pilot_level > p->minsignal)
  {
    if (p->lock_cnt < p->lock_delay)
      p->lock_cnt += (int)n;
  }
  else
    p->lock_cnt = 0;
  return p->lock_cnt >= p->lock_delay;
}

/* ------------------------------------------------------------------------- */
/* cFirFilter  (FirFilter.cpp)                                                */
/* ------------------------------------------------------------------------- */
#define MAX_NUMCOEF 75 /* FirFilter.h:15 */
typedef struct
{
  float fs;
  unsigned ntaps;
  int state;
  float coef[MAX_NUMCOEF * 2];
  float icoef[MAX_NUMCOEF * 2];
  float qcoef[MAX_NUMCOEF * 2];
  float rz[MAX_NUMCOEF];
  cf32 cz[MAX_NUMCOEF];
} firf;

static float izero(float x) /* FirFilter.cpp:39-58 */
{
  float x2 = x / 2.0f;
  float sum = 1.0f, ds = 1.0f, di = 1.0f;
  float errorlimit = (float)1e-9;
  float tmp;
  do
  {
    tmp = x2 / di;
    tmp *= tmp;
    ds *= tmp;
    sum += ds;
    di = (float)((double)di + 1.0);
  } while (ds >= errorlimit * sum);
  return sum;
}

static void firf_finish(firf* f)
{
  for (unsigned n = 0; n < f->ntaps; ++n)
    f->coef[n + f->ntaps] = f->coef[n];
  for (unsigned n = 0; n < f->ntaps * 2; ++n)
  {
    f->icoef[n] = f->coef[n];
    f->qcoef[n] = f->coef[n];
  }
  for (unsigned i = 0; i < f->ntaps; i++)
  {
    f->rz[i] = 0;
    f->cz[i].re = f->cz[i].im = 0;
  }
  f->state = 0;
}

/* FirFilter.cpp:78-148 */
static int firf_init_lp(firf* f, unsigned numtaps, float scale, float astop, float fpass,
                        float fstop, float fs)
{
  float beta;
  f->fs = fs;
  float nfpass = fpass / fs;
  float nfstop = fstop / fs;
  float nfcut = (nfstop + nfpass) / 2.0f;
  if (astop < 20.96f)
    beta = 0;
  else if (astop >= 50.0f)
    beta = (float)(.1102 * (double)(astop - 8.71f));
  else
    beta = (float)(.5842 * (double)powf(astop - 20.96f, (float)0.4) +
                   (double)(.07886f * (astop - 20.96f)));
  /* :101 double expression truncated to unsigned */
  f->ntaps = (unsigned)((double)(astop - 8.0f) /
                            ((double)2.285f * K_2PI * (double)(nfstop - nfpass)) +
                        1);
  if (f->ntaps > MAX_NUMCOEF)
    f->ntaps = MAX_NUMCOEF;
  if (f->ntaps < 3)
    f->ntaps = 3;
  if (numtaps)
    f->ntaps = numtaps;

  float fcenter = (float)(.5 * (double)(float)(f->ntaps - 1));
  float izb = izero(beta);
  for (unsigned n = 0; n < f->ntaps; ++n)
  {
    float x = (float)n - fcenter;
    float c;
    if ((float)n == fcenter)
      c = (float)(2.0 * (double)nfcut);
    else
      c = (float)((double)sinf((float)(K_2PI * (double)x * (double)nfcut)) / (K_PI * (double)x));
    x = ((float)n - ((float)f->ntaps - 1.0f) / 2.0f) / (((float)f->ntaps - 1.0f) / 2.0f);
    f->coef[n] = scale * c * izero(beta * sqrtf(1 - (x * x))) / izb;
  }
  firf_finish(f);
  return (int)f->ntaps;
}

unsigned fmo_design_lp_kaiser(float scale, float astop, float fpass, float fstop, float fs,
                              float* out, unsigned cap)
{
  firf f;
  firf_init_lp(&f, 0, scale, astop, fpass, fstop, fs);
  for (unsigned i = 0; i < f.ntaps && i < cap; i++)
    out[i] = f.coef[i];
  return f.ntaps;
}

/* FirFilter.cpp:302-320 */
static void firf_init_const(firf* f, unsigned numtaps, const float* c, float fs)
{
  f->fs = fs;
  f->ntaps = numtaps > MAX_NUMCOEF ? MAX_NUMCOEF : numtaps;
  for (unsigned i = 0; i < f->ntaps; ++i)
  {
    f->coef[i] = c[i];
    f->coef[f->ntaps + i] = c[i];
  }
  /* note: the reference leaves icoef/qcoef untouched here; only the real
   * Process() overload (which reads coef) is used with this init. */
  for (unsigned i = 0; i < f->ntaps; ++i)
  {
    f->rz[i] = 0;
    f->cz[i].re = f->cz[i].im = 0;
  }
  f->state = 0;
}

/* FirFilter.cpp:330-350, complex in place */
static void firf_process_cplx(firf* f, cf32* buf, unsigned n)
{
  for (unsigned i = 0; i < n; ++i)
  {
    f->cz[f->state] = buf[i];
    const float* hi = f->icoef + f->ntaps - f->state;
    const float* hq = f->qcoef + f->ntaps - f->state;
    float ar = hi[0] * f->cz[0].re;
    float ai = hq[0] * f->cz[0].im;
    for (unsigned j = 1; j < f->ntaps; j++)
    {
      ar += hi[j] * f->cz[j].re;
      ai += hq[j] * f->cz[j].im;
    }
    if (--f->state < 0)
      f->state += (int)f->ntaps;
    buf[i].re = ar;
    buf[i].im = ai;
  }
}

/* FirFilter.cpp:360-377, real in place */
static void firf_process_real(firf* f, float* buf, unsigned n)
{
  for (unsigned i = 0; i < n; ++i)
  {
    f->rz[f->state] = buf[i];
    const float* h = &f->coef[f->ntaps - f->state];
    float acc = h[0] * f->rz[0];
    for (unsigned j = 1; j < f->ntaps; ++j)
      acc += h[j] * f->rz[j];
    if (--f->state < 0)
      f->state += (int)f->ntaps;
    buf[i] = acc;
  }
}

/* FirFilter.cpp:387-413, two real streams sharing the complex delay line */
static void firf_process_two(firf* f, float* a, float* b, unsigned n)
{
  for (unsigned i = 0; i < n; ++i)
  {
    f->cz[f->state].re = a[i];
    f->cz[f->state].im = b[i];
    const float* hi = f->icoef + f->ntaps - f->state;
    const float* hq = f->qcoef + f->ntaps - f->state;
    float va = hi[0] * f->cz[0].re;
    float vb = hq[0] * f->cz[0].im;
    for (unsigned j = 1; j < f->ntaps; ++j)
    {
      va += hi[j] * f->cz[j].re;
      vb += hq[j] * f->cz[j].im;
    }
    if (--f->state < 0)
      f->state += (int)f->ntaps;
    a[i] = va;
    b[i] = vb;
  }
}

/* ------------------------------------------------------------------------- */
/* cIirFilter  (IirFilter.cpp)                                                */
/* ------------------------------------------------------------------------- */
enum
{
  FT_LP,
  FT_HP,
  FT_BP,
  FT_BR
};
typedef struct
{
  float a1, a2, b0, b1, b2;
  float w1a, w2a, w1b, w2b;
} iirf;

static void iirf_init(iirf* f, int type, float f0, float q, float fs)
{
  float w0 = (float)(K_2PI * (double)f0 / (double)fs);            /* :15 */
  float alpha = (float)((double)sinf(w0) / (2.0 * (double)q));    /* :16 */
  float A = (float)(1.0 / (1.0 + (double)alpha));                 /* :17 */
  double cw = (double)cosf(w0);
  switch (type)
  {
    case FT_LP:
      f->b0 = (float)((double)A * ((1.0 - cw) / 2.0));
      f->b1 = (float)((double)A * (1.0 - cw));
      f->b2 = (float)((double)A * ((1.0 - cw) / 2.0));
      f->a1 = (float)((double)A * (-2.0 * cw));
      f->a2 = (float)((double)A * (1.0 - (double)alpha));
      break;
    case FT_HP:
      f->b0 = (float)((double)A * ((1.0 + cw) / 2.0));
      f->b1 = (float)((double)(-A) * (1.0 + cw));
      f->b2 = (float)((double)A * ((1.0 + cw) / 2.0));
      f->a1 = (float)((double)A * (-2.0 * cw));
      f->a2 = (float)((double)A * (1.0 - (double)alpha));
      break;
    case FT_BP:
      f->b0 = A * alpha; /* :36 float*float */
      f->b1 = 0.0f;
      f->b2 = A * -alpha;
      f->a1 = (float)((double)A * (-2.0 * cw));
      f->a2 = (float)((double)A * (1.0 - (double)alpha));
      break;
    default: /* FT_BR :42-48 */
      f->b0 = (float)((double)A * 1.0);
      f->b1 = (float)((double)A * (-2.0 * cw));
      f->b2 = (float)((double)A * 1.0);
      f->a1 = (float)((double)A * (-2.0 * cw));
      f->a2 = (float)((double)A * (1.0 - (double)alpha));
      break;
  }
  f->w1a = f->w2a = f->w1b = f->w2b = 0;
}

void fmo_design_biquad(int type, float f0, float q, float fs, float o[5])
{
  iirf f;
  iirf_init(&f, type, f0, q, fs);
  o[0] = f.b0;
  o[1] = f.b1;
  o[2] = f.b2;
  o[3] = f.a1;
  o[4] = f.a2;
}

static void iirf_process_real(iirf* f, float* buf, unsigned n) /* :78-87 */
{
  for (unsigned i = 0; i < n; ++i)
  {
    float w0 = buf[i] - f->a1 * f->w1a - f->a2 * f->w2a;
    buf[i] = f->b0 * w0 + f->b1 * f->w1a + f->b2 * f->w2a;
    f->w2a = f->w1a;
    f->w1a = w0;
  }
}

static void iirf_process_two(iirf* f, float* a, float* b, unsigned n) /* :89-105 */
{
  for (unsigned i = 0; i < n; ++i)
  {
    float w0a = a[i] - f->a1 * f->w1a - f->a2 * f->w2a;
    float w0b = b[i] - f->a1 * f->w1b - f->a2 * f->w2b;
    a[i] = f->b0 * w0a + f->b1 * f->w1a + f->b2 * f->w2a;
    b[i] = f->b0 * w0b + f->b1 * f->w1b + f->b2 * f->w2b;
    f->w2a = f->w1a;
    f->w1a = w0a;
    f->w2b = f->w1b;
    f->w1b = w0b;
  }
}

/* ------------------------------------------------------------------------- */
/* CRDSDownConvert  (DownConvert.cpp:273-550, filtercoef.h)                   */
/* ------------------------------------------------------------------------- */
/* Half-band prototypes from filtercoef.h:62-150 (CuteSDR, Moe Wheatley, BSD).
 * Stored as the distinct non-zero side taps h[0], h[2], ... up to the one next
 * to the 0.5 centre tap; the full symmetric L-tap filter is rebuilt below. */
static const double HB_SIDE_11[] = {0.0060431029837374152, -0.049372515458761493,
                                    0.29332944952052842};
static const double HB_SIDE_15[] = {-0.001442203300285281, 0.013017512802724852,
                                    -0.061653278604903369, 0.30007792316024057};
static const double HB_SIDE_19[] = {0.00042366527106480427, -0.0040717333369021894,
                                    0.019895653881950692, -0.070740034412329067,
                                    0.30449249772844139};
static const double HB_SIDE_23[] = {-0.00014987651418332164, 0.0014748633283609852,
                                    -0.0074416944990005314,  0.026163522731980929,
                                    -0.077593699116544707,   0.30754683719791986};
static const double HB_SIDE_27[] = {0.000063730426952664685, -0.00061985193978569082,
                                    0.0031512504783365756,   -0.011173151342856621,
                                    0.03171888754393197,     -0.082917863582770729,
                                    0.3097770473566307};
static const double HB_SIDE_31[] = {-0.000030957335326552226, 0.00029271992847303054,
                                    -0.0014770381124258423,   0.0052539088990950535,
                                    -0.014856378748476874,    0.036406651919555999,
                                    -0.08699862567952929,     0.31140967076042625};
static const double HB_SIDE_35[] = {0.000017017718072971716, -0.00015425042851962818,
                                    0.00076219685751140838,  -0.002691614694785393,
                                    0.0075927497927344764,   -0.018325727896057686,
                                    0.040351004914363969,    -0.090198224668969554,
                                    0.31264689763504327};
static const double HB_SIDE_39[] = {-0.000010175082832074367, 0.000088036416015024345,
                                    -0.00042370835558387595,  0.0014772557414459019,
                                    -0.0041468438954260153,   0.0099579126901608011,
                                    -0.021433527104289002,    0.043598963493432855,
                                    -0.092695953625928404,    0.31358799113382152};
static const double HB_SIDE_43[] = {0.0000067666739082756387, -0.000055275221547958285,
                                    0.00025654074579418561,   -0.0008748125689163153,
                                    0.0024249876017061502,    -0.0057775190656021748,
                                    0.012299834239523121,     -0.024244050662087069,
                                    0.046354303503099069,     -0.094729903598633314,
                                    0.31433918020123208};
static const double HB_SIDE_47[] = {-0.0000045298314172004251, 0.000035333704512843228,
                                    -0.00015934776420643447,   0.0005340788063118928,
                                    -0.0014667949695500761,    0.0034792089350833247,
                                    -0.0073794356720317733,    0.014393786384683398,
                                    -0.026586603160193314,     0.048538673667907428,
                                    -0.09629115286535718,      0.31490673428547367};
static const double HB_SIDE_51[] = {0.0000033359253688981639, -0.000024584155158361803,
                                    0.00010677777483317733,   -0.00034890723143173914,
                                    0.00094239127078189603,   -0.0022118302078923137,
                                    0.0046575030752162277,    -0.0090130973415220566,
                                    0.016383673864361164,     -0.028697281101743237,
                                    0.05043292242400841,      -0.097611898315791965,
                                    0.31538104435015801};

typedef struct
{
  int len;
  const double* side;
  double max_bw; /* normalised alias-free bandwidth, filtercoef.h:45-56 */
} hb_proto;

static const hb_proto HB_PROTOS[] = {
    {11, HB_SIDE_11, (.5 - .475)}, {15, HB_SIDE_15, (.5 - .451)}, {19, HB_SIDE_19, (.5 - .428)},
    {23, HB_SIDE_23, (.5 - .409)}, {27, HB_SIDE_27, (.5 - .392)}, {31, HB_SIDE_31, (.5 - .378)},
    {35, HB_SIDE_35, (.5 - .366)}, {39, HB_SIDE_39, (.5 - .356)}, {43, HB_SIDE_43, (.5 - .347)},
    {47, HB_SIDE_47, (.5 - .340)}, {51, HB_SIDE_51, (.5 - .333)}};
#define N_HB_PROTOS 11
#define CIC3_MAX (.5 - .4985)
#define MIN_OUTPUT_RATE (7900.0 * 2.0) /* DownConvert.cpp:265 */
#define MAX_DECSTAGES 10
#define HB_BUFSIZE 32768 /* DownConvert.cpp:267 */

typedef struct
{
  int len;          /* 11 => the unrolled 11-tap class, 0 => CCicN3DecimateBy2, else generic */
  float coef[51];   /* RealType table (double literals narrowed to float) */
  cf32* buf;        /* generic: m_pHBFirBuf */
  cf32 d[10];       /* 11-tap: d0..d9 */
  cf32 xodd, xeven; /* CIC: m_Xodd, m_Xeven (DownConvert.cpp:692-696) */
} hb_stage;

typedef struct
{
  float out_rate, nco_freq, cw_offset, nco_inc, in_rate, max_bw;
  cf32 osc1;
  float osc_cos, osc_sin;
  int nstages;
  hb_stage st[MAX_DECSTAGES];
} rds_dc;

static void hb_stage_init(hb_stage* s, const hb_proto* p)
{
  memset(s, 0, sizeof(*s));
  s->len = p->len;
  int nside = (p->len + 1) / 4;
  for (int k = 0; k < nside; k++)
  {
    s->coef[2 * k] = (float)p->side[k];
    s->coef[p->len - 1 - 2 * k] = (float)p->side[k];
  }
  s->coef[(p->len - 1) / 2] = (float)0.5;
  if (p->len != 11)
    s->buf = (cf32*)calloc(HB_BUFSIZE, sizeof(cf32));
}

static void rdsdc_set_frequency(rds_dc* c, float nco_freq) /* DownConvert.cpp:311-320 */
{
  float tmpf = nco_freq + c->cw_offset;
  c->nco_freq = tmpf;
  c->nco_inc = (float)(K_2PI * (double)c->nco_freq / (double)c->in_rate);
  c->osc_cos = cosf(c->nco_inc);
  c->osc_sin = sinf(c->nco_inc);
}

/* (the CIC stage -- baseband >= 5.33 MHz, which cRadioReceiver never asks for, RadioReceiver.cpp:285 -- is
 * restated since round 6: cic3 below) */
static int rdsdc_init(rds_dc* c, float in_rate, float max_bw)
{
  memset(c, 0, sizeof(*c));
  c->osc1.re = 1.0f; /* :284 */
  c->osc1.im = 0.0f;
  c->in_rate = in_rate;
  c->max_bw = max_bw;
  float f = in_rate;
  int n = 0;
  /* :338-365 */
  while (((double)f > ((double)max_bw / HB_PROTOS[N_HB_PROTOS - 1].max_bw)) &&
         ((double)f > MIN_OUTPUT_RATE))
  {
    if (n >= MAX_DECSTAGES)
      return -1; /* (the reference's pointer array holds 10 stages, DownConvert.h; more would overrun it) */
    if ((double)f >= ((double)max_bw / CIC3_MAX))
    { /* :340-341 */
      memset(&c->st[n], 0, sizeof(c->st[n]));
      c->st[n++].len = 0;
      f = (float)((double)f / 2.0);
      continue;
    }
    for (int k = 0; k < N_HB_PROTOS; k++)
    {
      if ((double)f >= ((double)max_bw / HB_PROTOS[k].max_bw))
      {
        hb_stage_init(&c->st[n++], &HB_PROTOS[k]);
        break;
      }
    }
    f = (float)((double)f / 2.0);
  }
  c->nstages = n;
  c->out_rate = f;
  rdsdc_set_frequency(c, c->nco_freq); /* :368 */
  return 0;
}

static void rdsdc_free(rds_dc* c)
{
  for (int i = 0; i < c->nstages; i++)
    free(c->st[i].buf);
}

/* generic half-band, in place like the reference call at DownConvert.cpp:480
 * (pInData == pOutData); restates :512-550 incl. the double count of tap 0 */
static int hb_generic(hb_stage* s, int n, cf32* data)
{
  const int L = s->len;
  if (n < L)
    return n / 2; /* :519-520 unfiltered */
  for (int i = 0; i < n; i++)
    s->buf[L - 1 + i] = data[i];
  int nout = 0;
  const int mid = (L - 1) / 2;
  for (int i = 0; i < n; i += 2)
  {
    float ar = s->buf[i].re * s->coef[0];
    float ai = s->buf[i].im * s->coef[0];
    for (int j = 0; j < L; j += 2)
    {
      ar = ar + s->buf[i + j].re * s->coef[j];
      ai = ai + s->buf[i + j].im * s->coef[j];
    }
    ar = ar + s->buf[i + mid].re * s->coef[mid];
    ai = ai + s->buf[i + mid].im * s->coef[mid];
    data[nout].re = ar;
    data[nout].im = ai;
    nout++;
  }
  /* :546-547 reads the (already partly overwritten) in/out array */
  for (int i = 0, j = n - L + 1; i < L - 1; i++)
    s->buf[i] = data[j++];
  return nout;
}

/* 11-tap class (DownConvert.cpp:589-688): every output o uses window positions
 * 2o-10+{0,2,4,5,6,8,10} of [d0..d9 | input], summed left to right; floor(n/2)
 * outputs are returned; d = last 10 inputs.  In-place safe like the original. */
static int hb_11(hb_stage* s, int n, cf32* data)
{
  static const int T[7] = {0, 2, 4, 5, 6, 8, 10};
  cf32* tmp = (cf32*)malloc(sizeof(cf32) * (size_t)(n / 2 + 2));
  int nwritten = 9 + ((n - 11 - 6) / 2 > 0 ? (n - 11 - 6) / 2 : 0);
  if (nwritten > n / 2 + 1)
    nwritten = n / 2 + 1;
  for (int o = 0; o < nwritten; o++)
  {
    float ar = 0, ai = 0;
    for (int t = 0; t < 7; t++)
    {
      int w = 2 * o - 10 + T[t];
      cf32 x = (w < 0) ? s->d[10 + w] : data[w];
      float h = s->coef[T[t]];
      if (t == 0)
      {
        ar = h * x.re;
        ai = h * x.im;
      }
      else
      {
        ar = ar + h * x.re;
        ai = ai + h * x.im;
      }
    }
    tmp[o].re = ar;
    tmp[o].im = ai;
  }
  cf32 last[10];
  for (int k = 0; k < 10; k++)
    last[k] = data[n - 10 + k];
  for (int o = 0; o < nwritten; o++)
    data[o] = tmp[o];
  for (int k = 0; k < 10; k++)
    s->d[k] = last[k];
  free(tmp);
  return n / 2;
}

/* CCicN3DecimateBy2::DecBy2 (DownConvert.cpp:706-727), in place.  "InLength must be an even number" (:701): with
 * an odd one the loop reads pInData[InLength], whatever the buffer holds there -- restated as written (the buffer is
 * the caller's, stale contents included); the product refuses such calls. */
static int cic3(hb_stage* s, int n, cf32* data)
{
  int j = 0;
  for (int i = 0; i < n; i += 2, j++)
  { /* mag gn=8 */
    const cf32 even = data[i], odd = data[i + 1];
    /* .125 * (odd + m_Xeven + 3.0 * (m_Xodd + even)): float sums, then double arithmetic, narrowed on store */
    data[j].re = (float)(.125 * ((double)(odd.re + s->xeven.re) + 3.0 * (double)(s->xodd.re + even.re)));
    data[j].im = (float)(.125 * ((double)(odd.im + s->xeven.im) + 3.0 * (double)(s->xodd.im + even.im)));
    s->xodd = odd;
    s->xeven = even;
  }
  return j;
}

static int rdsdc_process(rds_dc* c, int n, cf32* data, cf32* out)
{
  /* quadrature-oscillator NCO with amplitude servo (:436-442, :464-465) */
  for (int i = 0; i < n; i++)
  {
    cf32 d = data[i];
    cf32 osc;
    osc.re = c->osc1.re * c->osc_cos - c->osc1.im * c->osc_sin;
    osc.im = c->osc1.im * c->osc_cos + c->osc1.re * c->osc_sin;
    float gn = (float)(1.95 - (double)(c->osc1.re * c->osc1.re + c->osc1.im * c->osc1.im));
    c->osc1.re = gn * osc.re;
    c->osc1.im = gn * osc.im;
    data[i].re = (d.re * osc.re) - (d.im * osc.im);
    data[i].im = (d.re * osc.im) + (d.im * osc.re);
  }
  int m = n;
  for (int k = 0; k < c->nstages; k++)
    m = (c->st[k].len == 0)    ? cic3(&c->st[k], m, data)
        : (c->st[k].len == 11) ? hb_11(&c->st[k], m, data)
                               : hb_generic(&c->st[k], m, data);
  for (int i = 0; i < m; i++)
    out[i] = data[i];
  return m;
}

/* ------------------------------------------------------------------------- */
/* cRDSGroupDecoder -> UECP frames  (RDSGroupDecoder.cpp)                     */
/* ------------------------------------------------------------------------- */
typedef struct
{
  uint8_t bytes[270];
  unsigned len;
} uecp_frame;

typedef struct
{
  /* members that Reset() does not touch start at zero (the
   * reference object is assumed to live in zeroed storage) */
  uint8_t seq_cnt;
  int stuff_ptr;
  uint8_t frame[263];
  int oda_map[32];
  int pty;
  int ta_tp;
  char ptyn[9];
  int ptyn_ab; /* bool in the reference */
  int ptyn_set;
  uint8_t di, di_prev;
  int di_finished;
  uint8_t ms, ms_prev;
  char ps_name[9];
  int ps_set;
  uint16_t pin;
  uint16_t pi_code;
  char rt_temp[66];
  int rt_first;
  int rt_ab;
  uint32_t rt_segreg;
  int rt_count;
  int rtp_template, rtp_scb, rtp_cb, rtp_rfu, rtp_ready;
  char ps_text[9]; /* function-static in the reference (RDSGroupDecoder.cpp:311) */

  uecp_frame* frames;
  unsigned nframes, capframes;
  char channel_name[9];
} group_dec;

static void gd_reset(group_dec* g) /* :136-164 */
{
  g->pi_code = 0;
  g->rt_segreg = 0;
  g->rt_count = 0;
  g->rt_first = 0;
  g->di = 0;
  g->di_prev = (uint8_t)-1;
  g->ms = 0;
  g->ms_prev = (uint8_t)-1;
  g->pin = (uint16_t)-1;
  g->ptyn_set = 0;
  g->ps_set = 0;
  g->ta_tp = -1;
  g->rtp_ready = 0;
  memset(g->rt_temp, 0, sizeof(g->rt_temp));
  memset(g->oda_map, 0, sizeof(g->oda_map));
  memset(g->ptyn, 0x20, sizeof(g->ptyn));
  memset(g->ps_name, 0x20, sizeof(g->ps_name));
}

static uint16_t crc16_ccitt(const uint8_t* p, int len) /* :961-977 */
{
  uint16_t crc = 0xffff;
  while (len--)
  {
    crc = (uint16_t)((crc >> 8) | (crc << 8));
    crc ^= *p++;
    crc ^= (uint16_t)((crc & 0xff) >> 4);
    crc ^= (uint16_t)((crc << 8) << 4);
    crc ^= (uint16_t)(((crc & 0xff) << 4) << 1);
  }
  return (uint16_t)~crc;
}

static void uecp_begin(group_dec* g) /* ClearUECPFrame :947-959 */
{
  g->frame[0] = 0;
  g->frame[1] = 0;
  g->frame[2] = g->seq_cnt;
  g->frame[3] = 0;
  g->stuff_ptr = 0;
}

static void uecp_put(group_dec* g, uint8_t v) /* AddStuffingValue :993-1001 */
{
  if (g->stuff_ptr > 255)
    return;
  g->frame[4 + g->stuff_ptr++] = v;
}

static void uecp_send(group_dec* g) /* SendUECPFrame :979-991, IsSettingActive()==false */
{
  g->seq_cnt++;
  g->frame[3] = (uint8_t)g->stuff_ptr;
  uint16_t crc = crc16_ccitt(g->frame, g->stuff_ptr + 4);
  g->frame[4 + g->stuff_ptr + 0] = (crc >> 8) & 0xff;
  g->frame[4 + g->stuff_ptr + 1] = crc & 0xff;
  if (g->nframes == g->capframes)
  {
    g->capframes = g->capframes ? g->capframes * 2 : 64;
    g->frames = (uecp_frame*)realloc(g->frames, g->capframes * sizeof(uecp_frame));
  }
  uecp_frame* f = &g->frames[g->nframes++];
  f->len = (unsigned)g->stuff_ptr + 6;
  memcpy(f->bytes, g->frame, f->len);
}

static void uecp_simple(group_dec* g, uint8_t mec, const uint8_t* payload, int n)
{
  uecp_begin(g);
  uecp_put(g, mec);
  uecp_put(g, 0x00);
  uecp_put(g, 0x01);
  for (int i = 0; i < n; i++)
    uecp_put(g, payload[i]);
  uecp_send(g);
}

static void gd_type0(group_dec* g, const uint16_t* b) /* :309-422 */
{
  int ctrl = (b[1] & 0x04) != 0;
  unsigned seg = b[1] & 0x03;
  uint8_t bit = (uint8_t)(1u << (3 - seg)); /* seg 3 -> d0 ... seg 0 -> d3 */
  if (ctrl)
    g->di |= bit;
  else if (g->di & bit)
    g->di ^= bit;
  g->di_finished++;

  int ta_tp = (b[1] & 0x10) ? 1 : 0;
  ta_tp |= (b[1] & 0x400) ? 2 : 0;
  if (g->ta_tp != ta_tp)
  {
    g->ta_tp = ta_tp;
    uint8_t v = (uint8_t)ta_tp;
    uecp_simple(g, 0x03, &v, 1);
  }
  if (g->di_finished >= 4 && g->di_prev != g->di)
  {
    g->di_finished = 0;
    g->di_prev = g->di;
    uint8_t v = g->di & 0xf;
    uecp_simple(g, 0x04, &v, 1);
  }
  g->ms = (b[1] & 0x08) ? 1 : 0;
  if (g->ms_prev != g->ms)
  {
    g->ms_prev = g->ms;
    uecp_simple(g, 0x05, &g->ms, 1);
  }
  g->ps_text[seg * 2] = (char)((b[3] >> 8) & 0xff);
  g->ps_text[seg * 2 + 1] = (char)(b[3] & 0xff);
  g->ps_set |= 1 << seg;
  if (g->ps_set == 0x0F)
  {
    if (memcmp(g->ps_name, g->ps_text, 8) != 0)
    {
      /* SetChannelName(ps_text) returns true when no dialog is open */
      memcpy(g->channel_name, g->ps_text, 8);
      g->channel_name[8] = 0;
      uecp_simple(g, 0x02, (const uint8_t*)g->ps_text, 8);
      memcpy(g->ps_name, g->ps_text, 8);
      g->ps_set = 0;
    }
  }
}

static void gd_type1(group_dec* g, const uint16_t* b, int version_b) /* :554-588 */
{
  if (g->pin != b[3])
  {
    g->pin = b[3];
    uint8_t v[2] = {(uint8_t)((g->pin >> 8) & 0xff), (uint8_t)(g->pin & 0xff)};
    uecp_simple(g, 0x06, v, 2);
  }
  if (!version_b)
  {
    uecp_begin(g);
    uecp_put(g, 0x1A);
    uecp_put(g, 0x00);
    uecp_put(g, (b[2] >> 8) & 0x7F);
    uecp_put(g, b[2] & 0xff);
    uecp_send(g);
  }
}

static void gd_type2(group_dec* g, const uint16_t* b, int version_b) /* :593-659 */
{
  unsigned ptr = b[1] & 0x0f;
  g->rtp_ready = 0;
  if (ptr == 0 && g->rt_first && g->rt_count > 1)
  {
    int ready = 1;
    for (int i = 0; i < g->rt_count; i++)
    {
      if (!(g->rt_segreg & (1u << i)))
      {
        ready = 0;
        g->rt_segreg = 0;
        g->rt_count = 0;
        break;
      }
    }
    if (ready)
    {
      uecp_begin(g);
      uecp_put(g, 0x0A);
      uecp_put(g, 0x00);
      uecp_put(g, 0x01);
      uecp_put(g, 65);
      uecp_put(g, (uint8_t)g->rt_ab);
      for (int i = 0; i < 64; i++)
        uecp_put(g, (uint8_t)g->rt_temp[i]);
      uecp_send(g);
      g->rtp_ready = 1;
    }
  }
  int ab = (b[1] >> 4) & 0x01;
  if (g->rt_ab != ab)
  {
    memset(g->rt_temp, 0x20, 66);
    g->rt_ab = ab;
    g->rt_first = 0;
    g->rt_segreg = 0;
    g->rt_count = 0;
  }
  if (!version_b)
  {
    g->rt_temp[ptr * 4] = (char)((b[2] >> 8) & 0xff);
    g->rt_temp[ptr * 4 + 1] = (char)(b[2] & 0xff);
    g->rt_temp[ptr * 4 + 2] = (char)((b[3] >> 8) & 0xff);
    g->rt_temp[ptr * 4 + 3] = (char)(b[3] & 0xff);
  }
  else
  {
    g->rt_temp[ptr * 2] = (char)((b[3] >> 8) & 0xff);
    g->rt_temp[ptr * 2 + 1] = (char)(b[3] & 0xff);
  }
  g->rt_segreg |= 1u << ptr;
  g->rt_count++;
  if (!g->rt_first && ptr == 0)
    g->rt_first = 1;
}

static void gd_type3a(group_dec* g, const uint16_t* b) /* :664-704 */
{
  int aid = b[3];
  uecp_begin(g);
  uecp_put(g, 0x40);
  uecp_put(g, b[1] & 0x1F);
  uecp_put(g, (b[3] >> 8) & 0xFF);
  uecp_put(g, b[3] & 0xFF);
  uecp_put(g, 0);
  uecp_put(g, (b[2] >> 8) & 0xFF);
  uecp_put(g, b[2] & 0xFF);
  uecp_put(g, 0);
  uecp_send(g);
  if (aid == 0x4bd7)
  {
    g->oda_map[b[1] & 0x1F] = 0x4bd7;
    g->rtp_template = b[2] & 0xFF;
    g->rtp_scb = (b[2] >> 8) & 0xF;
    g->rtp_cb = (b[2] >> 12) & 0x1;
    g->rtp_rfu = (b[2] >> 13) & 0x7;
  }
  else if (aid == 0xcd46)
    g->oda_map[b[1] & 0x1F] = 0xcd46;
  else
    g->oda_map[b[1] & 0x1F] = 0;
}

static void gd_type4a(group_dec* g, const uint16_t* b) /* :709-737 */
{
  double mjd = (double)(((b[1] & 0x03) << 15) | ((b[2] >> 1) & 0x7fff));
  unsigned hours = ((b[2] & 0x01u) << 4) | ((b[3] >> 12) & 0x0f);
  unsigned minutes = (b[3] >> 6) & 0x3f;
  int offset = b[3] & 0x3f;
  unsigned year = (unsigned)(int)((mjd - 15078.2) / 365.25);
  unsigned month = (unsigned)(int)((mjd - 14956.1 - (int)(year * 365.25)) / 30.6001);
  unsigned day = (unsigned)(mjd - 14956 - (int)(year * 365.25) - (int)(month * 30.6001));
  int K = ((month == 14) || (month == 15)) ? 1 : 0;
  year += (unsigned)(K + 1900);
  month -= (unsigned)(1 + K * 12);
  uecp_begin(g);
  uecp_put(g, 0x0D);
  uecp_put(g, (uint8_t)(year % 100));
  uecp_put(g, (uint8_t)month);
  uecp_put(g, (uint8_t)day);
  uecp_put(g, (uint8_t)hours);
  uecp_put(g, (uint8_t)minutes);
  uecp_put(g, 0);
  uecp_put(g, 0);
  uecp_put(g, (uint8_t)offset);
  uecp_send(g);
}

static void gd_type8a(group_dec* g, const uint16_t* b) /* :780-794 */
{
  uecp_begin(g);
  uecp_put(g, 0x30);
  uecp_put(g, 6);
  uecp_put(g, 0);
  uecp_put(g, b[1] & 0x1F);
  uecp_put(g, (b[2] >> 8) & 0xFF);
  uecp_put(g, b[2] & 0xFF);
  uecp_put(g, (b[3] >> 8) & 0xFF);
  uecp_put(g, b[3] & 0xFF);
  uecp_send(g);
}

static void gd_type10a(group_dec* g, const uint16_t* b) /* :812-844 */
{
  unsigned ptr = b[1] & 0x01;
  int ab = (b[1] >> 4) & 0x01;
  if (g->ptyn_ab != ab)
  {
    memset(g->ptyn, 0x20, 8);
    g->ptyn_ab = ab;
    g->ptyn_set = 0;
  }
  g->ptyn[ptr * 4] = (char)((b[2] >> 8) & 0xff);
  g->ptyn[ptr * 4 + 1] = (char)(b[2] & 0xff);
  g->ptyn[ptr * 4 + 2] = (char)((b[3] >> 8) & 0xff);
  g->ptyn[ptr * 4 + 3] = (char)(b[3] & 0xff);
  g->ptyn_set |= 1 << ptr;
  if (g->ptyn_set & 3)
    uecp_simple(g, 0x3A, (const uint8_t*)g->ptyn, 8);
}

static void gd_oda(group_dec* g, const uint16_t* b, int fn) /* :906-945 */
{
  if (fn == 0x4bd7)
  {
    if (g->rtp_ready)
    {
      uecp_begin(g);
      uecp_put(g, 0x46);
      uecp_put(g, 8);
      uecp_put(g, 0x4b);
      uecp_put(g, 0xd7);
      uecp_put(g, (b[1] >> 8) & 0xFF);
      uecp_put(g, b[1] & 0xFF);
      uecp_put(g, (b[2] >> 8) & 0xFF);
      uecp_put(g, b[2] & 0xFF);
      uecp_put(g, (b[3] >> 8) & 0xFF);
      uecp_put(g, b[3] & 0xFF);
      uecp_send(g);
      g->rtp_ready = 0;
    }
  }
  else if (fn == 0xcd46)
  {
    uecp_begin(g);
    uecp_put(g, 0x46);
    uecp_put(g, 7);
    uecp_put(g, 0xcd);
    uecp_put(g, 0x46);
    uecp_put(g, b[1] & 0xFF);
    uecp_put(g, (b[2] >> 8) & 0xFF);
    uecp_put(g, b[2] & 0xFF);
    uecp_put(g, (b[3] >> 8) & 0xFF);
    uecp_put(g, b[3] & 0xFF);
    uecp_send(g);
  }
}

static void gd_decode(group_dec* g, const uint16_t* b) /* DecodeRDS :166-269 */
{
  unsigned gt = (b[1] >> 11) & 0x1F;
  int version_b = (b[1] >> 11) & 0x1;
  uint16_t pi = b[0];
  if (pi != g->pi_code)
  { /* Decode_PI :271-286 */
    gd_reset(g);
    g->pi_code = pi;
    uint8_t v[2] = {(uint8_t)(pi & 0xff), (uint8_t)((pi >> 8) & 0xff)};
    uecp_simple(g, 0x01, v, 2);
  }
  int pty = (b[1] >> 5) & 0x1F;
  if (pty != g->pty)
  { /* Decode_PTY :288-303 */
    g->pty = pty;
    uint8_t v = (uint8_t)pty;
    uecp_simple(g, 0x07, &v, 1);
  }
  switch (gt)
  {
    case 0x00:
    case 0x01:
      gd_type0(g, b);
      break;
    case 0x02:
    case 0x03:
      gd_type1(g, b, version_b);
      break;
    case 0x04:
    case 0x05:
      gd_type2(g, b, version_b);
      break;
    case 0x06:
      gd_type3a(g, b);
      break;
    case 0x08:
      gd_type4a(g, b);
      break;
    case 0x10: /* 8A */
      if (g->oda_map[gt] > 0)
        gd_oda(g, b, g->oda_map[gt]);
      else
        gd_type8a(g, b);
      break;
    case 0x14:
      gd_type10a(g, b);
      break;
    case 0x1C: /* 14A/B, 15A, 15B: decoders are empty in the reference */
    case 0x1D:
    case 0x1E:
    case 0x1F:
      break;
    default: /* 5A..7A, 9A, 13A (empty non-ODA decoders) and the ODA-only types */
      if (g->oda_map[gt] > 0)
        gd_oda(g, b, g->oda_map[gt]);
      break;
  }
}

/* ------------------------------------------------------------------------- */
/* cRDSRxSignalProcessor  (RDSProcess.cpp)                                    */
/* ------------------------------------------------------------------------- */
#define RDS_FREQUENCY 57000.0
#define RDS_BITRATE (RDS_FREQUENCY / 48.0)
#define RDSPLL_RANGE 12.0
#define RDSPLL_BW 1.00
#define RDSPLL_ZETA 0.707
#define NUMBITS_CRC 10
#define NUMBITS_MSG 16
#define NUMBITS_BLOCK 26
#define BLOCK_ERROR_LIMIT 0
#define CRC_POLY 0x5B9
#define GROUPB_BIT 0x0800
enum
{
  ST_BITSYNC,
  ST_BLOCKSYNC,
  ST_GROUPDECODE,
  ST_GROUPRESYNC
};
static const uint32_t OFFSET_SYN[8] = {0x3D8, 0x3D4, 0x25C, 0x258, 0x3D8, 0x3D4, 0x3CC, 0x258};
/* parity-check matrix rows for the 16 message bits (RDS standard, RDSProcess.cpp:24-41) */
static const uint32_t PARCKH[16] = {0x2DC, 0x16E, 0x0B7, 0x287, 0x39F, 0x313, 0x355, 0x376,
                                    0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B};

typedef struct
{
  uint16_t blocks[4];
  unsigned call_index;
} group_rec;

typedef struct
{
  float sample_rate, process_rate;
  rds_dc dc;
  cf32* arr_in;
  cf32* raw;
  float* mag;
  float* data;
  float* match_coef;
  unsigned match_len;
  float last_sync, last_sync_slope, last_data;
  float nco_phase, nco_freq, nco_llimit, nco_hlimit, pll_alpha, pll_beta;
  firf lpf, matched;
  iirf bitsync;
  int last_bit;
  uint32_t in_bits;
  int cur_block, cur_bitpos, state, bgroup_off, block_errors;
  uint16_t block_data[4];
  group_dec gd;
  group_rec* groups;
  unsigned ngroups, capgroups;
  unsigned call_index;
  unsigned last_len;
  /* taps */
  cf32* tap_lpf;
  float* tap_pll;
  float* tap_mf;
  float* tap_sync;
} rds_proc;

static void rds_reset(rds_proc* r) /* RDSProcess.cpp:92-118 */
{
  gd_reset(&r->gd);
  r->nco_phase = 0.0f;
  r->nco_freq = 0.0f;
  firf_init_lp(&r->lpf, 0, 1.0f, 40.0f, 2400.0f, (float)(1.3 * 2400.0), r->process_rate);
  firf_init_const(&r->matched, r->match_len, r->match_coef, r->process_rate);
  iirf_init(&r->bitsync, FT_BP, (float)RDS_BITRATE, 500, r->process_rate);
  r->last_sync = 0;
  r->last_sync_slope = 0;
  r->last_bit = 0;
  r->cur_bitpos = 0;
  r->cur_block = 0;
  r->state = ST_BITSYNC;
  r->bgroup_off = 0;
  r->last_data = 0;
}

static int rds_init(rds_proc* r, float sample_rate) /* :43-81 */
{
  memset(r, 0, sizeof(*r));
  r->sample_rate = sample_rate;
  if (rdsdc_init(&r->dc, sample_rate, 8000.0f) < 0)
    return -1;
  r->process_rate = r->dc.out_rate;
  rdsdc_set_frequency(&r->dc, (float)(-RDS_FREQUENCY));
  r->arr_in = (cf32*)calloc(FMO_MAX_BLOCK, sizeof(cf32));
  r->raw = (cf32*)calloc(FMO_MAX_BLOCK, sizeof(cf32));
  r->mag = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  r->data = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  r->tap_lpf = (cf32*)calloc(FMO_MAX_BLOCK, sizeof(cf32));
  r->tap_pll = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  r->tap_mf = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  r->tap_sync = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));

  float norm = (float)(K_2PI / (double)r->process_rate);
  /* m_RdsNcoFreq holds its default initialiser 0.0 here (RDSProcess.h:86) */
  r->nco_llimit = (float)(((double)0.0f - RDSPLL_RANGE) * (double)norm);
  r->nco_hlimit = (float)(((double)0.0f + RDSPLL_RANGE) * (double)norm);
  r->pll_alpha = (float)(2.0 * RDSPLL_ZETA * RDSPLL_BW * (double)norm);
  r->pll_beta = (float)((double)(r->pll_alpha * r->pll_alpha) / (4.0 * RDSPLL_ZETA * RDSPLL_ZETA));

  unsigned L = (unsigned)((double)r->process_rate / RDS_BITRATE); /* :65 */
  r->match_coef = (float*)calloc(L * 2 + 1, sizeof(float));
  for (int i = 0; i <= (int)L; i++)
  {
    float t = (float)i / r->process_rate;
    float x = (float)((double)t * RDS_BITRATE);
    float x64 = (float)(64.0 * (double)x);
    double shape = (1.0 / (1.0 / (double)x - (double)x64)) - (1.0 / (9.0 / (double)x - (double)x64));
    double c = (double)cosf((float)(2.0 * K_2PI * (double)x));
    r->match_coef[(unsigned)i + L] = (float)(.75 * c * shape);
    r->match_coef[L - (unsigned)i] = (float)(-.75 * c * shape);
  }
  r->match_len = L * 2; /* :77: the last of the 2L+1 values is never used */
  rds_reset(r);
  return 0;
}

static void rds_free(rds_proc* r)
{
  rdsdc_free(&r->dc);
  free(r->arr_in);
  free(r->raw);
  free(r->mag);
  free(r->data);
  free(r->match_coef);
  free(r->groups);
  free(r->gd.frames);
  free(r->tap_lpf);
  free(r->tap_pll);
  free(r->tap_mf);
  free(r->tap_sync);
}

/* RDSProcess.cpp:187-217 -- polynomial arctan with double intermediates */
static inline float rds_arctan2(float y, float x)
{
  if (x == 0.0f)
  {
    if (y > 0.0f)
      return (float)K_PI2;
    if (y == 0.0f)
      return 0.0f;
    return (float)-K_PI2;
  }
  float angle;
  float z = y / x;
  if (fabsf(z) < 1.0f)
  {
    angle = (float)((double)z / (1.0 + 0.2854 * (double)z * (double)z));
    if (x < 0.0f)
    {
      if (y < 0.0f)
        return (float)((double)angle - K_PI);
      return (float)((double)angle + K_PI);
    }
  }
  else
  {
    angle = (float)(K_PI2 - (double)z / ((double)(z * z) + 0.2854));
    if (y < 0.0f)
      return (float)((double)angle - K_PI);
  }
  return angle;
}

float fmo_rds_arctan2(float y, float x)
{
  return rds_arctan2(y, x);
}

/* cRtlSdrSource::ReadAsyncCB, RTL_SDR_Source.cpp:207-211: ComplexType(buf[2i] / (255.0/2.0) - 1.0,
 * buf[2i+1] / (255.0/2.0) - 1.0) -- double arithmetic, narrowed by the complex<float> ctor. */
void fmo_convert_u8(const uint8_t* buf, unsigned samples, float* iq)
{
  for (unsigned i = 0; i < samples; i++)
  {
    iq[2 * i] = (float)(buf[2 * i] / (255.0 / 2.0) - 1.0);
    iq[2 * i + 1] = (float)(buf[2 * i + 1] / (255.0 / 2.0) - 1.0);
  }
}

static void rds_pll(rds_proc* r, const cf32* in, float* out, unsigned n) /* :222-270 */
{
  for (unsigned i = 0; i < n; i++)
  {
    float s, c;
    sincos_nco(r->nco_phase, &s, &c);
    float tr = c * in[i].re - s * in[i].im;
    float ti = c * in[i].im + s * in[i].re;
    float err = -rds_arctan2(ti, tr);
    r->nco_freq += (r->pll_beta * err);
    if (r->nco_freq > r->nco_hlimit)
      r->nco_freq = r->nco_hlimit;
    else if (r->nco_freq < r->nco_llimit)
      r->nco_freq = r->nco_llimit;
    r->nco_phase += (r->nco_freq + r->pll_alpha * err);
    out[i] = ti;
  }
  r->nco_phase = fmodf(r->nco_phase, (float)K_2PI); /* :269 */
}

static uint32_t rds_check_block(rds_proc* r, uint32_t offset, int use_fec) /* :377-431 */
{
  uint32_t tb = 0x3FFFFFF & r->in_bits;
  uint32_t syn = tb >> 16;
  for (int i = 0; i < NUMBITS_MSG; i++)
  {
    if (tb & 0x8000)
      syn ^= PARCKH[i];
    tb <<= 1;
  }
  syn ^= offset;
  if (syn && use_fec)
  {
    uint32_t mask = 1u << (NUMBITS_BLOCK - 1);
    for (int i = 0; i < NUMBITS_MSG; i++)
    {
      if (syn & 0x200)
      {
        if ((syn & 0x1F) == 0)
        {
          r->in_bits ^= mask;
          syn <<= 1;
        }
        else
        {
          syn <<= 1;
          syn ^= CRC_POLY;
        }
      }
      else
        syn <<= 1;
      mask >>= 1;
    }
    syn &= 0x3FF;
  }
  return syn;
}

static void rds_emit_group(rds_proc* r)
{
  if (r->ngroups == r->capgroups)
  {
    r->capgroups = r->capgroups ? r->capgroups * 2 : 64;
    r->groups = (group_rec*)realloc(r->groups, r->capgroups * sizeof(group_rec));
  }
  memcpy(r->groups[r->ngroups].blocks, r->block_data, sizeof(r->block_data));
  r->groups[r->ngroups].call_index = r->call_index;
  r->ngroups++;
  gd_decode(&r->gd, r->block_data);
}

static void rds_store_block(rds_proc* r)
{
  r->block_data[r->cur_block] = (uint16_t)(r->in_bits >> NUMBITS_CRC);
  if (r->cur_block == 1 && (r->block_data[1] & GROUPB_BIT))
    r->bgroup_off = 4;
  else
    r->bgroup_off = 0;
}

static void rds_new_bit(rds_proc* r, int bit) /* :272-375 */
{
  r->in_bits = (r->in_bits << 1) | (uint32_t)bit;
  switch (r->state)
  {
    case ST_BITSYNC:
      if (!rds_check_block(r, OFFSET_SYN[0], 0))
      {
        r->cur_bitpos = 0;
        r->bgroup_off = 0;
        r->block_data[0] = (uint16_t)(r->in_bits >> NUMBITS_CRC);
        r->cur_block = 1;
        r->state = ST_BLOCKSYNC;
      }
      break;
    case ST_BLOCKSYNC:
      if (++r->cur_bitpos < NUMBITS_BLOCK)
        break;
      r->cur_bitpos = 0;
      if (rds_check_block(r, OFFSET_SYN[r->cur_block + r->bgroup_off], 0))
        r->state = ST_BITSYNC;
      else
      {
        rds_store_block(r);
        if (r->cur_block >= 3)
        {
          r->cur_block = 0;
          r->block_errors = 0;
          r->state = ST_GROUPDECODE;
          rds_emit_group(r);
        }
        else
          r->cur_block++;
      }
      break;
    case ST_GROUPDECODE:
      if (++r->cur_bitpos < NUMBITS_BLOCK)
        break;
      r->cur_bitpos = 0;
      if (rds_check_block(r, OFFSET_SYN[r->cur_block + r->bgroup_off], 1))
      {
        r->block_errors++;
        if (r->block_errors > BLOCK_ERROR_LIMIT)
          r->state = ST_BITSYNC;
        else
        {
          if (++r->cur_block > 3)
            r->cur_block = 0;
          if (r->cur_block != 0)
            r->state = ST_GROUPRESYNC;
        }
      }
      else
      {
        rds_store_block(r);
        if (++r->cur_block > 3)
        {
          r->cur_block = 0;
          r->block_errors = 0;
          rds_emit_group(r);
        }
      }
      break;
    case ST_GROUPRESYNC:
      if (++r->cur_bitpos < NUMBITS_BLOCK)
        break;
      r->cur_bitpos = 0;
      if (++r->cur_block > 3)
      {
        r->cur_block = 0;
        r->state = ST_GROUPDECODE;
      }
      break;
  }
}

static void rds_process(rds_proc* r, const float* in, unsigned n) /* :120-180 */
{
  for (unsigned i = 0; i < n; i++)
  {
    r->arr_in[i].re = in[i];
    r->arr_in[i].im = 0.0f;
  }
  unsigned len = (unsigned)rdsdc_process(&r->dc, (int)n, r->arr_in, r->raw);
  firf_process_cplx(&r->lpf, r->raw, len);
  memcpy(r->tap_lpf, r->raw, len * sizeof(cf32));
  rds_pll(r, r->raw, r->data, len);
  memcpy(r->tap_pll, r->data, len * sizeof(float));
  firf_process_real(&r->matched, r->data, len);
  memcpy(r->tap_mf, r->data, len * sizeof(float));
  for (unsigned i = 0; i < len; i++)
    r->mag[i] = r->data[i] * r->data[i];
  iirf_process_real(&r->bitsync, r->mag, len);
  memcpy(r->tap_sync, r->mag, len * sizeof(float));
  for (unsigned i = 0; i < len; i++)
  {
    float d = r->data[i];
    float sv = r->mag[i];
    float slope = sv - r->last_sync;
    r->last_sync = sv;
    if ((slope < 0.0f) && (r->last_sync_slope * slope) < 0.0f)
    {
      int bit = (r->last_data >= 0) ? 1 : 0;
      rds_new_bit(r, bit ^ r->last_bit);
      r->last_bit = bit;
    }
    r->last_data = d;
    r->last_sync_slope = slope;
  }
  r->last_len = len;
}

/* ------------------------------------------------------------------------- */
/* cFmDecoder  (FmDecode.cpp:237-539)                                         */
/* ------------------------------------------------------------------------- */
struct fmo_decoder
{
  float fs_if, fs_bb;
  int table_size, tuning_shift;
  float freq_dev;
  unsigned downsample;
  int stereo;
  float if_level, bb_mean, bb_level;
  float demod_gain;
  cf32* buf_tuned;
  cf32* buf_demod;
  float* buf_bb;
  float* buf_mono;
  float* buf_stereo;
  float* buf_raw;
  fine_tuner tuner;
  pilot_pll pilot;
  dsf rs_in, rs_mono, rs_stereo;
  rds_proc rds;
  iirf notch;
  firf lpf;
  float de_re, de_im, de_alpha;
  float nco_phase, nco_incr, nco_hl, nco_ll, pll_alpha, pll_beta, dc_off;
  /* taps */
  unsigned n_demod, n_audio;
  float* tap_raw;
  float* tap_mono;
  float* tap_stereo;
};

static void dec_reset(fmo_decoder* d) /* :326-338 */
{
  d->stereo = 0;
  d->if_level = 0;
  d->bb_mean = 0;
  d->bb_level = 0;
  d->dc_off = 0;
  d->nco_incr = 0.0f;
  d->nco_phase = 0.0f;
  rds_reset(&d->rds);
}

void fmo_reset(fmo_decoder* d)
{
  dec_reset(d);
}

fmo_decoder* fmo_create(const fmo_params* p)
{
  fmo_decoder* d = (fmo_decoder*)calloc(1, sizeof(*d));
  unsigned D = p->downsample ? p->downsample : 1;
  d->fs_if = (float)p->sample_rate_if;
  d->fs_bb = (float)(p->sample_rate_if / D); /* :248 double / unsigned, stored as float */
  d->table_size = p->table_size ? (int)p->table_size : 64;
  d->tuning_shift = p->use_shift_override
                        ? p->tuning_shift_override
                        : (int)lrint(-(double)d->table_size * p->tuning_offset / p->sample_rate_if);
  d->freq_dev = (float)60000.0;
  d->downsample = D;
  /* :254  1.0 / (60000.0 / float * K_2PI) */
  d->demod_gain = (float)(1.0 / (60000.0 / (double)d->fs_bb * K_2PI));
  fine_tuner_init(&d->tuner, (unsigned)d->table_size, d->tuning_shift);
  /* :257-260  freq: double/float -> float param; bandwidth: int/float */
  pilot_init(&d->pilot, (float)(19000.0 / (double)d->fs_bb), 50 / d->fs_bb, 0.04f);
  unsigned if_order = p->if_filter_order ? p->if_filter_order : 8 * D;
  dsf_init(&d->rs_in, if_order, 0.6 / D, (double)D, 1);
  unsigned ao = (unsigned)(int)((double)d->fs_bb / 1000.0);
  dsf_init(&d->rs_mono, ao, p->bandwidth_pcm / (double)d->fs_bb,
           (double)d->fs_bb / p->sample_rate_pcm, 0);
  dsf_init(&d->rs_stereo, ao, p->bandwidth_pcm / (double)d->fs_bb,
           (double)d->fs_bb / p->sample_rate_pcm, 0);
  if (rds_init(&d->rds, d->fs_bb) < 0)
  {
    free(d);
    return NULL;
  }
  d->buf_tuned = (cf32*)calloc(FMO_MAX_BLOCK, sizeof(cf32));
  d->buf_demod = (cf32*)calloc(FMO_MAX_BLOCK, sizeof(cf32));
  d->buf_bb = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->buf_mono = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->buf_stereo = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->buf_raw = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->tap_raw = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->tap_mono = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));
  d->tap_stereo = (float*)calloc(FMO_MAX_BLOCK, sizeof(float));

  iirf_init(&d->notch, FT_BR, (float)19000.0, 5, (float)p->sample_rate_pcm); /* :285 */
  firf_init_lp(&d->lpf, 0, 1.0f, 60.0f, 15000.0f, (float)(1.4 * 15000.0),
               (float)p->sample_rate_pcm); /* :286 */
  {
    /* InitDeemphasis :340-346 with Time = 50e-6 / 75e-6 narrowed to float */
    float tc = p->us_version ? (float)75E-6 : (float)50E-6;
    float sr = (float)p->sample_rate_pcm;
    d->de_alpha = (1.0f - expf(-1.0f / (sr * tc)));
    d->de_re = d->de_im = 0.0f;
  }
  {
    /* :305-312 */
    float fac = (float)(K_2PI / (double)d->fs_bb);
    float bandwidth = 0.85f * d->fs_bb;
    float maxdev = 0.95f * (0.5f * d->fs_bb);
    d->nco_ll = (-maxdev) * fac;
    d->nco_hl = (+maxdev) * fac;
    d->pll_alpha = 0.125f * bandwidth * fac;
    d->pll_beta = (d->pll_alpha * d->pll_alpha) / 2.0f;
  }
  dec_reset(d);
  return d;
}

void fmo_destroy(fmo_decoder* d)
{
  if (!d)
    return;
  free(d->tuner.table);
  dsf_free(&d->rs_in);
  dsf_free(&d->rs_mono);
  dsf_free(&d->rs_stereo);
  rds_free(&d->rds);
  free(d->buf_tuned);
  free(d->buf_demod);
  free(d->buf_bb);
  free(d->buf_mono);
  free(d->buf_stereo);
  free(d->buf_raw);
  free(d->tap_raw);
  free(d->tap_mono);
  free(d->tap_stereo);
  free(d);
}

#define DC_ALPHA 0.0001 /* FmDecode.cpp:361 (double) */
static void fm_pll(fmo_decoder* d, const cf32* sig, float* out, unsigned n) /* :362-415 */
{
  float dc = d->dc_off;
  for (unsigned i = 0; i < n; ++i)
  {
    float s, c;
    sincos_nco(d->nco_phase, &s, &c);
    /* ComplexType(Cos, Sin) * signal[i] */
    float re = c * sig[i].re - s * sig[i].im;
    float im = c * sig[i].im + s * sig[i].re;
    float err = -atan2f(im, re);
    d->nco_incr += d->pll_beta * err;
    if (d->nco_incr < d->nco_ll)
      d->nco_incr = d->nco_ll;
    if (d->nco_incr > d->nco_hl)
      d->nco_incr = d->nco_hl;
    d->nco_phase += d->nco_incr + d->pll_alpha * err;
    if ((double)d->nco_phase >= K_2PI)
      d->nco_phase = (float)fmod((double)d->nco_phase, K_2PI);
    while (d->nco_phase < 0)
      d->nco_phase = (float)((double)d->nco_phase + K_2PI);
    float pinc = 2 * d->nco_incr;
    dc = (float)((1 - DC_ALPHA) * (double)dc + DC_ALPHA * (double)pinc);
    out[i] = (pinc - dc) * d->demod_gain;
  }
  d->dc_off = dc;
}

unsigned fmo_process_stream(fmo_decoder* d, const float* iq, unsigned samples, float* audio)
{
  /* Preconditions the reference leaves unchecked: its work buffers hold 65536 samples
   * (FmDecode.cpp:277-282) and its half-band delay lines 32768 (DownConvert.cpp:267,500), so a
   * call whose baseband length exceeds 32768 - taps + 1 overruns the heap there.  Refuse. */
  if (samples == 0 || samples > FMO_MAX_BLOCK ||
      (samples + d->downsample - 1) / d->downsample + 51 > HB_BUFSIZE)
    return 0;
  unsigned n = samples;
  const cf32* in = (const cf32*)iq;
  d->rds.call_index++;

  fine_tuner_process(&d->tuner, in, d->buf_tuned, n); /* :424 */
  {
    /* RMSLevelApprox :505-519 */
    unsigned m = (n + 63) / 64;
    float level = 0;
    for (unsigned i = 0; i < m; ++i)
    {
      float re = d->buf_tuned[i].re, im = d->buf_tuned[i].im;
      level += re * re + im * im;
    }
    float rms = sqrtf(level / (float)m);
    d->if_level = 0.95f * d->if_level + 0.05f * rms; /* :427 */
  }
  n = dsf_process_cplx(&d->rs_in, d->buf_tuned, d->buf_demod, n); /* :430 */
  d->n_demod = n;
  fm_pll(d, d->buf_demod, d->buf_bb, n); /* :433 */
  rds_process(&d->rds, d->buf_bb, n);    /* :436 */
  {
    /* SamplesMeanRMS :522-539 */
    float vsum = 0, vsumsq = 0;
    for (unsigned i = 0; i < n; ++i)
    {
      float v = d->buf_bb[i];
      vsum += v;
      vsumsq += v * v;
    }
    float mean = vsum / (float)n;
    float rms = sqrtf(vsumsq / (float)n);
    d->bb_mean = 0.95f * d->bb_mean + 0.05f * mean;
    d->bb_level = 0.95f * d->bb_level + 0.05f * rms;
  }
  unsigned mono_n = dsf_process_real(&d->rs_mono, d->buf_bb, d->buf_mono, n); /* :445 */
  d->stereo = pilot_process(&d->pilot, d->buf_bb, d->buf_raw, n);             /* :448 */
  for (unsigned i = 0; i < n; ++i)
    d->buf_raw[i] *= 2 * d->buf_bb[i]; /* :455-456 */
  memcpy(d->tap_raw, d->buf_raw, n * sizeof(float));
  n = dsf_process_real(&d->rs_stereo, d->buf_raw, d->buf_stereo, n); /* :464 */
  (void)mono_n;
  d->n_audio = n;
  memcpy(d->tap_mono, d->buf_mono, n * sizeof(float));
  memcpy(d->tap_stereo, d->buf_stereo, n * sizeof(float));

  firf_process_two(&d->lpf, d->buf_stereo, d->buf_mono, n); /* :469 */
  for (unsigned i = 0; i < n; ++i)
  { /* ProcessDeemphasisFilter :348-359 */
    d->de_re = (1.0f - d->de_alpha) * d->de_re + d->de_alpha * d->buf_stereo[i];
    d->buf_stereo[i] = d->de_re * 2.0f;
    d->de_im = (1.0f - d->de_alpha) * d->de_im + d->de_alpha * d->buf_mono[i];
    d->buf_mono[i] = d->de_im * 2.0f;
  }
  iirf_process_two(&d->notch, d->buf_stereo, d->buf_mono, n); /* :471 */

  if (d->stereo)
  {
    for (unsigned i = 0; i < n; ++i)
    {
      float m = d->buf_mono[i], s = d->buf_stereo[i];
      audio[2 * i] = (m + s) * 0.5f;
      audio[2 * i + 1] = (m - s) * 0.5f;
    }
  }
  else
  {
    for (unsigned i = 0; i < n; ++i)
    {
      float m = d->buf_mono[i] * 0.5f;
      audio[2 * i] = m;
      audio[2 * i + 1] = m;
    }
  }
  return 2 * n;
}

void fmo_get_status(const fmo_decoder* d, fmo_status* st)
{
  st->stereo = d->stereo;
  /* FmDecode.h:146-150 */
  float tuned = (float)(-d->tuning_shift) * d->fs_if / (float)d->table_size;
  st->tuning_offset = tuned + d->bb_mean * d->freq_dev;
  st->if_level = d->if_level;
  st->baseband_level = d->bb_level;
  st->pilot_level = 2 * d->pilot.pilot_level; /* FmDecode.h:75 */
  st->rds_state = d->rds.state;
}

void fmo_get_taps(const fmo_decoder* d, fmo_taps* t)
{
  t->n_demod = d->n_demod;
  t->demod = (const float*)d->buf_demod;
  t->baseband = d->buf_bb;
  t->pilot38 = d->tap_raw;
  t->n_audio = d->n_audio;
  t->mono_rs = d->tap_mono;
  t->stereo_rs = d->tap_stereo;
  t->n_rds = d->rds.last_len;
  t->rds_lpf = (const float*)d->rds.tap_lpf;
  t->rds_pll = d->rds.tap_pll;
  t->rds_mf = d->rds.tap_mf;
  t->rds_sync = d->rds.tap_sync;
}

unsigned fmo_rds_group_count(const fmo_decoder* d)
{
  return d->rds.ngroups;
}

void fmo_rds_group_get(const fmo_decoder* d, unsigned idx, uint16_t blocks[4], unsigned* call_index)
{
  memcpy(blocks, d->rds.groups[idx].blocks, 8);
  *call_index = d->rds.groups[idx].call_index;
}

unsigned fmo_uecp_frame_count(const fmo_decoder* d)
{
  return d->rds.gd.nframes;
}

unsigned fmo_uecp_frame_get(const fmo_decoder* d, unsigned idx, uint8_t* out, unsigned cap)
{
  const uecp_frame* f = &d->rds.gd.frames[idx];
  unsigned n = f->len < cap ? f->len : cap;
  memcpy(out, f->bytes, n);
  return f->len;
}

void fmo_debug_push_group(fmo_decoder* d, const uint16_t blocks[4])
{
  gd_decode(&d->rds.gd, blocks);
}

const char* fmo_channel_name(const fmo_decoder* d)
{
  return d->rds.gd.channel_name;
}

static unsigned copy_out(const float* src, unsigned n, float* out, unsigned cap)
{
  for (unsigned i = 0; i < n && i < cap; i++)
    out[i] = src[i];
  return n;
}

unsigned fmo_design_tuner_lut(unsigned table_size, int freq_shift, float* out, unsigned cap)
{
  fine_tuner ft;
  fine_tuner_init(&ft, table_size, freq_shift);
  unsigned n = copy_out((const float*)ft.table, 2 * table_size, out, cap);
  free(ft.table);
  return n;
}
unsigned fmo_get_lut(const fmo_decoder* d, float* out, unsigned cap)
{
  return copy_out((const float*)d->tuner.table, 2 * d->tuner.size, out, cap);
}
unsigned fmo_get_if_taps(const fmo_decoder* d, float* out, unsigned cap)
{
  return copy_out(d->rs_in.coeff, d->rs_in.order + 2, out, cap);
}
unsigned fmo_get_audio_taps(const fmo_decoder* d, float* out, unsigned cap)
{
  return copy_out(d->lpf.coef, d->lpf.ntaps, out, cap);
}
unsigned fmo_get_rds_lpf_taps(const fmo_decoder* d, float* out, unsigned cap)
{
  return copy_out(d->rds.lpf.coef, d->rds.lpf.ntaps, out, cap);
}
unsigned fmo_get_rds_mf_taps(const fmo_decoder* d, float* out, unsigned cap)
{
  return copy_out(d->rds.matched.coef, d->rds.matched.ntaps, out, cap);
}
unsigned fmo_get_rds_hb_lengths(const fmo_decoder* d, int* out, unsigned cap)
{
  for (int i = 0; i < d->rds.dc.nstages && (unsigned)i < cap; i++)
    out[i] = d->rds.dc.st[i].len;
  return (unsigned)d->rds.dc.nstages;
}

unsigned fmo_get_constants(const fmo_decoder* d, double* o, unsigned cap)
{
  double v[] = {
      (double)d->tuning_shift,      /* 0 */
      d->demod_gain,                /* 1 */
      d->de_alpha,                  /* 2 */
      d->pll_alpha,                 /* 3 */
      d->pll_beta,                  /* 4 */
      d->nco_hl,                    /* 5 */
      d->nco_ll,                    /* 6 */
      d->pilot.minfreq,             /* 7 */
      d->pilot.maxfreq,             /* 8 */
      d->pilot.b0,                  /* 9 */
      d->pilot.a1,                  /* 10 */
      d->pilot.a2,                  /* 11 */
      d->pilot.lf_b0,               /* 12 */
      d->pilot.lf_b1,               /* 13 */
      d->pilot.freq,                /* 14: current pilot NCO frequency */
      (double)d->pilot.lock_delay,  /* 15 */
      (double)d->rs_mono.order,     /* 16 */
      (double)(float)d->rs_mono.downsample, /* 17 */
      d->rds.process_rate,          /* 18 */
      d->rds.dc.nco_inc,            /* 19 */
      d->rds.dc.osc_cos,            /* 20 */
      d->rds.dc.osc_sin,            /* 21 */
      d->rds.pll_alpha,             /* 22 */
      d->rds.pll_beta,              /* 23 */
      d->rds.nco_hlimit,            /* 24 */
      d->rds.nco_llimit,            /* 25 */
      d->fs_bb,                     /* 26 */
      (double)d->rds.match_len,     /* 27 */
  };
  unsigned n = sizeof(v) / sizeof(v[0]);
  for (unsigned i = 0; i < n && i < cap; i++)
    o[i] = v[i];
  return n;
}


/* ============================================================================================ */
/* cRadioReceiver: the stream members around the decoder (RadioReceiver.cpp, see fmd_oracle.h)   */
/* ============================================================================================ */
#include <ctype.h>
#include <stdio.h>

#define STREAM_TIME_BASE 1000000 /* Kodi timing_constants.h (third-party header, not vendored) */
#define OUTPUT_SAMPLERATE 48000  /* Definitions.h:15 */
#define DEMUX_SPECIALID_STREAMCHANGE (-11)

typedef struct rx_block
{
  float* iq;
  unsigned samples;
  struct rx_block* next;
} rx_block;

struct fmo_receiver
{
  fmo_decoder* dec;      /* m_FMDecoder */
  double if_rate;        /* m_IfRate */
  double tuner_freq;     /* m_activeTunerFreq */
  char adapter_name[128];
  char channel_name[16]; /* m_channelName */
  int stream_change;     /* m_StreamChange */
  float audio_level;     /* m_AudioLevel (RadioReceiver.h:105: float, 0.0f) */
  float audio_mean, audio_rms;
  double pts_next;       /* m_PTSNext */
  uint64_t queued;       /* m_AudioSourceSize */
  rx_block *head, *tail; /* m_AudioSourceBuffer */
  int end_marked, buffer_warning;
  uint8_t* uecp;         /* m_UECPOutputBuffer */
  size_t uecp_len, uecp_cap;
  unsigned frames_seen;  /* frames of the decoder log already stuffed */
  char name_seen[16];
  uint8_t* packet;
  size_t packet_cap;
};

fmo_receiver* fmo_receiver_open(const fmo_params* p, double tuner_freq, const char* adapter_name)
{
  fmo_receiver* r = (fmo_receiver*)calloc(1, sizeof(*r));
  r->dec = fmo_create(p); /* :296-300 */
  if (!r->dec)
  {
    free(r);
    return NULL;
  }
  r->if_rate = p->sample_rate_if;
  r->tuner_freq = tuner_freq;
  snprintf(r->adapter_name, sizeof(r->adapter_name), "%s", adapter_name ? adapter_name : "");
  r->stream_change = 1;               /* :345 */
  r->pts_next = STREAM_TIME_BASE;     /* :347 */
  fmo_reset(r->dec);                  /* :349 */
  return r;
}

void fmo_receiver_close(fmo_receiver* r)
{
  if (!r)
    return;
  while (r->head)
  {
    rx_block* b = r->head;
    r->head = b->next;
    free(b->iq);
    free(b);
  }
  fmo_destroy(r->dec);
  free(r->uecp);
  free(r->packet);
  free(r);
}

void fmo_receiver_write(fmo_receiver* r, const float* iq, unsigned samples) /* :426-436 */
{
  if (!samples)
    return;
  rx_block* b = (rx_block*)calloc(1, sizeof(*b));
  b->iq = (float*)malloc((size_t)samples * 2 * sizeof(float));
  memcpy(b->iq, iq, (size_t)samples * 2 * sizeof(float));
  b->samples = samples;
  r->queued += samples;
  if (r->tail)
    r->tail->next = b;
  else
    r->head = b;
  r->tail = b;
}

void fmo_receiver_write_u8(fmo_receiver* r, const uint8_t* buf, unsigned samples)
{ /* RTL_SDR_Source.cpp:196-213: convert, then WriteDataBuffer */
  float* tmp = (float*)malloc((size_t)samples * 2 * sizeof(float));
  fmo_convert_u8(buf, samples, tmp);
  fmo_receiver_write(r, tmp, samples);
  free(tmp);
}

void fmo_receiver_end(fmo_receiver* r) /* :438-443 */
{
  r->end_marked = 1;
}

uint64_t fmo_receiver_queued_samples(const fmo_receiver* r) /* :420-424 */
{
  return r->queued;
}

void fmo_receiver_set_stream_change(fmo_receiver* r) /* RadioReceiver.h:83 */
{
  r->stream_change = 1;
}

static void rx_push(fmo_receiver* r, uint8_t v)
{
  if (r->uecp_len == r->uecp_cap)
  {
    r->uecp_cap = r->uecp_cap ? 2 * r->uecp_cap : 1024;
    r->uecp = (uint8_t*)realloc(r->uecp, r->uecp_cap);
  }
  r->uecp[r->uecp_len++] = v;
}

/* cRadioReceiver::AddUECPDataFrame, :387-414 */
static int rx_add_uecp_frame(fmo_receiver* r, const uint8_t* frame, unsigned length)
{
  if (r->uecp_len > 16384)
    return 0;
  rx_push(r, 0xFE);
  for (unsigned i = 0; i < length; i++)
  {
    uint8_t value = frame[i];
    if (value < 0xFD)
      rx_push(r, value);
    else
    {
      rx_push(r, 0xFD);
      rx_push(r, (uint8_t)((value & 3) - 1));
    }
  }
  rx_push(r, 0xFF);
  return 1;
}

/* cRadioReceiver::SamplesMeanRMS, :584-598 (float sums, float quotient, float sqrt) */
static void rx_samples_mean_rms(const float* samples, double* mean, double* rms, unsigned n)
{
  float vsum = 0;
  float vsumsq = 0;
  for (unsigned i = 0; i < n; ++i)
  {
    float v = samples[i];
    vsum += v;
    vsumsq += v * v;
  }
  *mean = vsum / n;
  *rms = sqrtf(vsumsq / n);
}

/* the upward calls of one ProcessStream, replayed in order: frames go through AddUECPDataFrame,
 * a new PS name through SetChannelName (:600-612: m_channelName = Trim(name)) */
static void rx_collect_callbacks(fmo_receiver* r)
{
  unsigned n = fmo_uecp_frame_count(r->dec);
  for (; r->frames_seen < n; r->frames_seen++)
  {
    uint8_t f[300];
    unsigned len = fmo_uecp_frame_get(r->dec, r->frames_seen, f, sizeof(f));
    rx_add_uecp_frame(r, f, len);
  }
  const char* name = fmo_channel_name(r->dec);
  if (name[0] && strcmp(name, r->name_seen) != 0)
  {
    snprintf(r->name_seen, sizeof(r->name_seen), "%s", name);
    const char* a = name;
    while (*a && isspace((unsigned char)*a))
      a++;
    size_t e = strlen(a);
    while (e && isspace((unsigned char)a[e - 1]))
      e--;
    memcpy(r->channel_name, a, e);
    r->channel_name[e] = 0;
  }
}

int fmo_receiver_demux_read(fmo_receiver* r, fmo_packet* pkt) /* :462-542 */
{
  memset(pkt, 0, sizeof(*pkt));
  if (r->stream_change)
  { /* :471-477 */
    pkt->stream_id = DEMUX_SPECIALID_STREAMCHANGE;
    r->stream_change = 0;
    return 1;
  }
  if (r->uecp_len)
  { /* :482-503 */
    if (r->packet_cap < r->uecp_len)
    {
      r->packet_cap = r->uecp_len;
      r->packet = (uint8_t*)realloc(r->packet, r->packet_cap);
    }
    memcpy(r->packet, r->uecp, r->uecp_len);
    pkt->data = r->packet;
    pkt->stream_id = 2;
    pkt->size = (int)r->uecp_len;
    pkt->pts = r->pts_next;
    r->uecp_len = 0;
    return 1;
  }
  if (!r->buffer_warning && (double)r->queued > 10 * r->if_rate) /* :510-514 */
    r->buffer_warning = 1;
  if (!r->head)
    return r->end_marked ? 0 : -1; /* :447-459 */
  rx_block* b = r->head;
  r->head = b->next;
  if (!r->head)
    r->tail = NULL;
  r->queued -= b->samples;
  size_t need = (size_t)b->samples * sizeof(float) * 2; /* :519-520 */
  if (r->packet_cap < need)
  {
    r->packet_cap = need;
    r->packet = (uint8_t*)realloc(r->packet, r->packet_cap);
  }
  unsigned iSize = fmo_process_stream(r->dec, b->iq, b->samples, (float*)r->packet); /* :524-525 */
  free(b->iq);
  free(b);
  rx_collect_callbacks(r);

  double audio_mean, audio_rms;
  rx_samples_mean_rms((const float*)r->packet, &audio_mean, &audio_rms, iSize); /* :527 */
  r->audio_mean = (float)audio_mean;
  r->audio_rms = (float)audio_rms;
  r->audio_level = (float)(0.95 * r->audio_level + 0.05 * audio_rms); /* :528 */

  double duration = (double)(iSize)*STREAM_TIME_BASE / 2 / OUTPUT_SAMPLERATE; /* :531 */
  pkt->data = r->packet;
  pkt->stream_id = 1;
  pkt->size = (int)(iSize * sizeof(float));
  pkt->duration = duration;
  pkt->pts = r->pts_next;
  r->pts_next = r->pts_next + duration; /* :538 */
  return 1;
}

/* GetSignalStatus(float&, float&, bool&), :544-556.  log10 of a float argument resolves to the
 * float overload under libstdc++; 20 * float is float, + 3.01 is double. */
int fmo_receiver_signal_status(fmo_receiver* r, float* interface_db, float* audio_db, int* stereo)
{
  if (!r->dec || r->stream_change)
    return 0;
  fmo_status st;
  fmo_get_status(r->dec, &st);
  *interface_db = 20 * log10f(st.if_level);
  *audio_db = (float)(20 * log10f(r->audio_level) + 3.01);
  *stereo = st.stereo;
  return 1;
}

/* GetSignalStatus(int, PVRSignalStatus&), :558-582.  The format string consumes five of its six
 * arguments (IF= prints the tuned frequency, BB= the interface level, Audio= the baseband level);
 * restated as written. */
int fmo_receiver_pvr_signal_status(fmo_receiver* r, char* adapter_name, unsigned name_cap,
                                   char* adapter_status, unsigned status_cap, char* provider_name,
                                   unsigned provider_cap, int* signal, int* snr)
{
  if (!r->dec || r->stream_change)
    return 0;
  fmo_status st;
  fmo_get_status(r->dec, &st);
  float interfaceLevel = 20 * log10f(st.if_level);
  float audioLevel = (float)(20 * log10f(r->audio_level) + 3.01);
  snprintf(adapter_status, status_cap, "Freq.=%8.4fMHz - %s - IF=%+5.1fdB  BB=%+5.1fdB  Audio=%+5.1fdB",
           r->tuner_freq / 1000000, st.stereo ? "Stereo" : "Mono",
           (r->tuner_freq + st.tuning_offset) * 1.0e-6, interfaceLevel,
           20 * log10f(st.baseband_level) + 3.01);
  snprintf(adapter_name, name_cap, "%s", r->adapter_name);
  snprintf(provider_name, provider_cap, "%s", r->channel_name);
  *signal = (int)(2.5 * (interfaceLevel + 40) * 656);
  *snr = (int)((audioLevel + 100) * 656);
  return 1;
}

void fmo_receiver_audio_level(const fmo_receiver* r, float* mean, float* rms, float* level)
{
  *mean = r->audio_mean;
  *rms = r->audio_rms;
  *level = r->audio_level;
}

fmo_decoder* fmo_receiver_decoder(fmo_receiver* r)
{
  return r->dec;
}
