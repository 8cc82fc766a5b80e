/* Host sweep of the transcendental restatements in csrc/fmd_math.h against the host libm /
 * x87 (the functions the reference's CPU build calls).  Prints mismatch counts. */
#include <stdio.h>
#include <stdlib.h>
#include "../../pvr.rtl.radiofm_amd/csrc/fmd_math.h"

static inline void x87(float p, float* s, float* c)
{
  float sv, cv;
  __asm__ volatile("fsincos" : "=t"(cv), "=u"(sv) : "0"(p));
  *s = sv;
  *c = cv;
}
static uint64_t st = 88172645463325252ull;
static inline uint64_t rnd(void)
{
  st ^= st << 13;
  st ^= st >> 7;
  st ^= st << 17;
  return st;
}

int main(int argc, char** argv)
{
  long n = argc > 1 ? atol(argv[1]) : 10000000L;
  static double tab[2 * FMD_SINCOS_TAB_SIZE];
  float atab[FMD_ATAN_TAB_FLOATS];
  fmd_atan_table_fill(atab);
  const long double twopi = 6.283185307179586476925286766559005768L;
  for (int k = 0; k < FMD_SINCOS_TAB_SIZE; k++)
  {
    long double a = twopi * k / (long double)FMD_SINCOS_TAB_SIZE;
    tab[2 * k] = (double)sinl(a);
    tab[2 * k + 1] = (double)cosl(a);
  }
  struct FmdSincosTab t;
  {
    long double h = twopi / (long double)FMD_SINCOS_TAB_SIZE;
    t.inv_h = (double)(1.0L / h);
    double hd = (double)h;
    uint64_t u;
    memcpy(&u, &hd, 8);
    u &= ~((1ull << 15) - 1);
    memcpy(&hd, &u, 8);
    t.h_hi = hd;
    t.h_lo = (double)(h - (long double)hd);
  }
  static double tab256[2 * FMD_SINCOS_P256_SIZE];
  for (int k = 0; k < FMD_SINCOS_P256_SIZE; k++)
  {
    tab256[2 * k] = (double)sinl((long double)k / 256.0L);
    tab256[2 * k + 1] = (double)cosl((long double)k / 256.0L);
  }
  long bad_a = 0, bad_f = 0, bad_s = 0, bad_t = 0, bad_r = 0, bad_p = 0;
  for (long i = 0; i < n; i++)
  {
    float y, x;
    uint64_t r = rnd();
    switch (i & 3)
    {
      case 0:
        y = fmd_u2f((uint32_t)r);
        x = fmd_u2f((uint32_t)(r >> 32));
        if (y != y || x != x)
          continue;
        break;
      case 1:
        y = ((int32_t)(r & 0xffffffff)) / 2147483648.0f * 2.0f;
        x = ((int32_t)(r >> 32)) / 2147483648.0f * 2.0f;
        break;
      case 2:
        y = ((int32_t)(r & 0xffffffff)) / 2147483648.0f * 1e-3f;
        x = ((int32_t)(r >> 32)) / 2147483648.0f;
        break;
      default:
        y = ((int32_t)(r & 0xffffffff)) / 2147483648.0f;
        x = ((int32_t)(r >> 32)) / 2147483648.0f * 3e-2f;
    }
    const float ref = atan2f(y, x);
    const float a = fmd_atan2f(y, x), f = fmd_atan2f_tab(y, x, atab);
    if (fmd_f2u(ref) != fmd_f2u(a) && !(ref != ref && a != a))
      bad_a++;
    if (fmd_f2u(ref) != fmd_f2u(f) && !(ref != ref && f != f))
      bad_f++;
    const float p = (float)(((rnd() >> 11) * (1.0 / 9007199254740992.0) - 0.3) * 20.0);
    float s1, c1, s2, c2, s3, c3;
    x87(p, &s1, &c1);
    fmd_sincos_nco(p, &s2, &c2);
    fmd_sincos_tab(p, tab, t, &s3, &c3);
    bad_s += (fmd_f2u(s1) != fmd_f2u(s2)) + (fmd_f2u(c1) != fmd_f2u(c2));
    bad_t += (fmd_f2u(s1) != fmd_f2u(s3)) + (fmd_f2u(c1) != fmd_f2u(c3));
    (void)bad_r;
    /* the exact-reduction form of the serial stage's NCOs: phases in [0, 2 pi] */
    const float p2 = (float)((rnd() >> 11) * (1.0 / 9007199254740992.0) * 6.2832);
    x87(p2, &s1, &c1);
    fmd_sincos_p256(p2, tab256, &s3, &c3);
    bad_p += (fmd_f2u(s1) != fmd_f2u(s3)) + (fmd_f2u(c1) != fmd_f2u(c3));
  }
  /* RTL-SDR byte -> float (RTL_SDR_Source.cpp:207-211): all 256 inputs */
  long bad_u = 0;
  for (unsigned b = 0; b < 256; b++)
  {
    const float ref = (float)(b / (255.0 / 2.0) - 1.0);
    bad_u += fmd_f2u(ref) != fmd_f2u(fmd_u8_to_f32(b));
  }
  printf("n=%ld atan2f=%ld atan2f_tab=%ld sincos_nco=%ld sincos_tab=%ld u8_to_f32=%ld sincos_p256=%ld\n", n,
         bad_a, bad_f, bad_s, bad_t, bad_u, bad_p);
  return 0;
}
