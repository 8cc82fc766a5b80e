/*
 * fmd_math.h -- transcendental helpers for the HIP kernels, written so that their float
 * results equal what the reference's CPU build produces:
 *
 *  - fmd_atan2f() / fmd_atan2f_tab(): the reference calls atan2(float,float) -> glibc atan2f
 *    (FmDecode.cpp:395).  glibc 2.35 (this image; libm is a third-party dependency of the
 *    reference, not vendored) implements atan2f/atanf with the fdlibm float algorithms
 *    (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c; Sun Microsystems 1993).  The published
 *    algorithm is restated here operation for operation in float arithmetic; with
 *    -ffp-contract=off every IEEE add/mul/div rounds like the host's, so the result is
 *    bit-identical (tests/test_device_math_cpu.py sweeps both forms against the host libm).
 *  - fmd_sincos_nco() / fmd_sincos_tab(): the reference uses the x87 fsincos instruction on
 *    the float phase and stores the 64-bit-mantissa result to float (FmDecode.cpp:167,386,
 *    RDSProcess.cpp:245).  That is the correctly rounded float sin/cos except for
 *    double-rounding cases of probability ~2^-40.  Here: evaluate in FP64 (< 1 ulp of
 *    double), round once to float; disagreement probability ~2^-28 per call (0 observed in
 *    4x10^8).
 *  - fmd_rds_arctan2(): the reference's own polynomial arctan (RDSProcess.cpp:187-217),
 *    float with double intermediates.
 *
 *  - fmd_u8_to_f32(): the RTL-SDR byte -> float conversion (RTL_SDR_Source.cpp:207-211),
 *    float(b / (255.0 / 2.0) - 1.0) in double.  The real value is (2b-255)/255; a float
 *    rounding boundary is a dyadic rational and (2b-255)/255 stays >= 2^-40 away from every
 *    one of them, far more than the 2^-53 the two double roundings can move it, so the result
 *    is the correctly rounded float of 2b/255 - 1.  Computed as b*c_hi - 1 (exact: c_hi has 9
 *    significant bits) plus b*c_lo with one rounding, c_hi + c_lo = 2/255 to 2^-48; all 256
 *    inputs are checked in the CPU tests.
 *
 * Usable from host C (CPU sweep of the restatement) and from HIP device code.
 */
#ifndef FMD_MATH_H
#define FMD_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __HIPCC__
#define FMD_HD __host__ __device__ static inline
#define FMD_HD_NOINLINE __host__ __device__ static __attribute__((noinline))
#else
#define FMD_HD static inline
#define FMD_HD_NOINLINE static __attribute__((noinline))
#endif

#if defined(__clang__)
typedef float fmd_v2f __attribute__((ext_vector_type(2)));
#else
typedef float fmd_v2f __attribute__((vector_size(8)));
#endif

/* (a.x - b.y, a.y + b.x): one packed add with the second operand's halves swapped and the low one
 * negated (operand modifiers, no extra instruction) */
FMD_HD fmd_v2f fmd_pk_add_cross(fmd_v2f a, fmd_v2f b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  fmd_v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  fmd_v2f r = {a[0] - b[1], a[1] + b[0]};
  return r;
#endif
}

#define FMD_K_2PI (2.0 * 3.14159265358979323846)
#define FMD_K_PI (3.14159265358979323846)
#define FMD_K_PI2 (FMD_K_PI / 2.0)

/* Rare-input fix-ups in the per-sample loops sit behind a wave-uniform test (one compare and one
 * scalar branch on the common path) instead of a divergent branch (exec save / restore). */
#if defined(__HIP_DEVICE_COMPILE__)
#define FMD_ANY_LANE(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#else
#define FMD_ANY_LANE(cond) (cond)
#endif

/* n / d, correctly rounded, for operands where no step of the quotient refinement leaves the
 * normal range (here: d in [0.4, 2^26], |n| <= 2^26 or zero).  On the device this is the
 * compiler's own IEEE division expansion without the range scaling and special-case fix-up
 * around it (8 instead of 11 instructions); the fmaf calls are that algorithm, not contractions
 * of reference arithmetic. */
FMD_HD float fmd_div_midrange(float n, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = n * r1;
  const float e1 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e2, r1, q1);
#else
  return n / d;
#endif
}

FMD_HD uint32_t fmd_f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
FMD_HD float fmd_u2f(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* fdlibm s_atanf.c */
/* RTL_SDR_Source.cpp:207-211, see the header comment.  The fmaf calls are this function's own
 * evaluation scheme, not contractions of reference arithmetic. */
FMD_HD float fmd_u8_to_f32(unsigned b)
{
  const float f = (float)b;
  const float s = __builtin_fmaf(f, 0x1.01p-7f, -1.0f); /* exact */
  return __builtin_fmaf(f, 0x1.010102p-23f, s);
}

FMD_HD float fmd_atanf(float x)
{
  const float atanhi0 = 4.6364760399e-01f, atanhi1 = 7.8539812565e-01f,
              atanhi2 = 9.8279368877e-01f, atanhi3 = 1.5707962513e+00f;
  const float atanlo0 = 5.0121582440e-09f, atanlo1 = 3.7748947079e-08f,
              atanlo2 = 3.4473217170e-08f, atanlo3 = 7.5497894159e-08f;
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
              aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
              aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
              aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
  const float one = 1.0f;
  float w, s1, s2, z;
  int32_t hx = (int32_t)fmd_f2u(x);
  int32_t ix = hx & 0x7fffffff;
  int id;
  float hi = 0.0f, lo = 0.0f;
  if (ix >= 0x4c000000)
  { /* |x| >= 2^25 */
    if (ix > 0x7f800000)
      return x + x; /* NaN */
    if (hx > 0)
      return atanhi3 + atanlo3;
    return -atanhi3 - atanlo3;
  }
  if (ix < 0x3ee00000)
  { /* |x| < 0.4375 */
    if (ix < 0x31000000)
      return x; /* |x| < 2^-29 */
    id = -1;
  }
  else
  {
    x = fabsf(x);
    if (ix < 0x3f980000)
    { /* |x| < 1.1875 */
      if (ix < 0x3f300000)
      { /* 7/16 <= |x| < 11/16 */
        id = 0;
        x = (2.0f * x - one) / (2.0f + x);
        hi = atanhi0;
        lo = atanlo0;
      }
      else
      { /* 11/16 <= |x| < 19/16 */
        id = 1;
        x = (x - one) / (x + one);
        hi = atanhi1;
        lo = atanlo1;
      }
    }
    else
    {
      if (ix < 0x401c0000)
      { /* |x| < 2.4375 */
        id = 2;
        x = (x - 1.5f) / (one + 1.5f * x);
        hi = atanhi2;
        lo = atanlo2;
      }
      else
      { /* 2.4375 <= |x| < 2^25 */
        id = 3;
        x = -1.0f / x;
        hi = atanhi3;
        lo = atanlo3;
      }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0)
    return x - x * (s1 + s2);
  z = hi - ((x * (s1 + s2) - lo) - x);
  return (hx < 0) ? -z : z;
}

/* fdlibm e_atan2f.c */
FMD_HD_NOINLINE float fmd_atan2f(float y, float x)
{
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
              pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  float z;
  int32_t hx = (int32_t)fmd_f2u(x), hy = (int32_t)fmd_f2u(y);
  int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if ((ix > 0x7f800000) || (iy > 0x7f800000))
    return x + y; /* NaN */
  if (hx == 0x3f800000)
    return fmd_atanf(y); /* x = 1.0 */
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); /* 2*sign(x) + sign(y) */
  if (iy == 0)
  {
    switch (m)
    {
      case 0:
      case 1:
        return y;
      case 2:
        return pi + tiny;
      default:
        return -pi - tiny;
    }
  }
  if (ix == 0)
    return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000)
  {
    if (iy == 0x7f800000)
    {
      switch (m)
      {
        case 0:
          return pi_o_4 + tiny;
        case 1:
          return -pi_o_4 - tiny;
        case 2:
          return 3.0f * pi_o_4 + tiny;
        default:
          return -3.0f * pi_o_4 - tiny;
      }
    }
    else
    {
      switch (m)
      {
        case 0:
          return 0.0f;
        case 1:
          return -0.0f;
        case 2:
          return pi + tiny;
        default:
          return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000)
    return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  int32_t k = (iy - ix) >> 23;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo; /* |y/x| > 2^60 */
  else if (hx < 0 && k < -60)
    z = 0.0f; /* |y|/x < -2^60 */
  else
    z = fmd_atanf(fabsf(y / x));
  switch (m)
  {
    case 0:
      return z;
    case 1:
      return fmd_u2f(fmd_f2u(z) ^ 0x80000000u);
    case 2:
      return pi - (z - pi_lo);
    default:
      return (z - pi_lo) - pi;
  }
}

/* sin/cos of a float phase evaluated in double and rounded once to float.
 * Branch-free for the NCO range (|phase| < ~1e5): k = rint(x*2/pi), two-term Cody-Waite
 * reduction with fma, fdlibm __kernel_sin/__kernel_cos polynomials on [-pi/4, pi/4] (< 1 ulp
 * of double), quadrant fix-up by selects, then ONE rounding to float. */
FMD_HD void fmd_sincos_nco(float phase, float* s, float* c)
{
  const double x = (double)phase;
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  /* sin kernel */
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(r * z, ps, r);
  /* cos kernel */
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * z * pc);
  const int n = (int)k;
  double so = (n & 1) ? cr : sr;
  double co = (n & 1) ? sr : cr;
  so = (n & 2) ? -so : so;
  co = ((n + 1) & 2) ? -co : co;
  *s = (float)so;
  *c = (float)co;
}

/* Compact form of the same fdlibm atan2f for the per-sample PLL loop (no data-dependent branch
 * except the rare-input test).  Per reduced range r (index 0 = no reduction, 1..4 = fdlibm id
 * 0..3) the table holds a, b, c, d, hi, lo (the function takes hi and lo from it; a..d document
 * the rows, the function derives them from two packed constants, see there) with
 *   reduced argument = (a*q + b) / (c*q + d)        and   result = hi - ((p - lo) - xr)
 * which reproduces every fdlibm operation: a*q and c*q are exact or the reference's own products
 * (1*q, 2*q, 1.5*q, 0*q), adding b / d is the reference's add / subtract, and hi = lo = 0 turns
 * the final expression into xr - p, the unreduced branch.  A single unsigned range test on the
 * quotient sends zeros, infinities, NaNs and |y/x| outside [2^-29, 2^25) (which includes every
 * exponent gap beyond 60) to the literal fmd_atan2f().  Bit-identical to glibc 2.35 atan2f on
 * 6x10^8 inputs incl. range-edge stress (tests/test_device_math_cpu.py runs a shorter sweep). */
#define FMD_ATAN_TAB_FLOATS 40 /* 5 ranges x 8 floats (6 used) */
FMD_HD void fmd_atan_table_fill(float* t)
{
  const float v[5][8] = {
      {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0, 0},
      {2.0f, -1.0f, 1.0f, 2.0f, 4.6364760399e-01f, 5.0121582440e-09f, 0, 0},
      {1.0f, -1.0f, 1.0f, 1.0f, 7.8539812565e-01f, 3.7748947079e-08f, 0, 0},
      {1.0f, -1.5f, 1.5f, 1.0f, 9.8279368877e-01f, 3.4473217170e-08f, 0, 0},
      {0.0f, -1.0f, 1.0f, 0.0f, 1.5707962513e+00f, 7.5497894159e-08f, 0, 0}};
  for (int r = 0; r < 5; r++)
    for (int k = 0; k < 8; k++)
      t[r * 8 + k] = v[r][k];
}

#if defined(__HIP_DEVICE_COMPILE__)
/* two floats as the 64-bit scalar operand of a packed instruction */
FMD_HD unsigned long long fmd_pack2f(float lo, float hi)
{
  return (unsigned long long)fmd_f2u(lo) | ((unsigned long long)fmd_f2u(hi) << 32);
}
#endif

/* core: the common-range result and whether this lane needs the literal path instead */
/* rare_measure: iq - 0x31000000 as an unsigned number; the lane needs the literal function when it is
 * >= FMD_ATAN_RARE_LIMIT (a loop takes the maximum over a group of samples and tests once) */
#define FMD_ATAN_RARE_LIMIT (0x4c000000u - 0x31000000u)
FMD_HD float fmd_atan2f_tab_core(float y, float x, const float* tab, uint32_t* rare_measure)
{
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
              aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
              aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
              aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
  const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const float q = fabsf(y / x);
  const uint32_t iq = fmd_f2u(q);
  const uint32_t rare_m = iq - 0x31000000u; /* >= FMD_ATAN_RARE_LIMIT: see above */
  /* the common-range evaluation runs for every lane (a rare lane computes a value nobody uses:
   * its range index is clamped, nothing here can trap) */
  /* no clamp for the rare lanes: whatever their bits are, the range index below stays within 0..4,
   * nothing on the way can trap, and their result is replaced by the literal function's */
  const uint32_t ic = iq;
  const float qc = fmd_u2f(ic);
  /* range index without compares: (ic - T) is negative, i.e. shifts to -1, exactly when ic < T */
  const int r = 4 + ((int32_t)(ic - 0x3ee00000u) >> 31) + ((int32_t)(ic - 0x3f300000u) >> 31) +
                ((int32_t)(ic - 0x3f980000u) >> 31) + ((int32_t)(ic - 0x401c0000u) >> 31);
  const float* t = tab + 8 * r;
  /* a = d and c = -b in every row (a: 1 2 1 1 0, c: 0 1 1 1.5 1), so the reduction only needs two
   * small numbers per range; they come out of two packed nibble constants (a, 2c) with a bit-field
   * extract and a conversion instead of waiting for the table row, which is then only needed for
   * hi / lo at the very end. */
  /* Numerator and denominator are taken TWICE as large, which leaves the quotient as it is (a power
   * of two scales every product, sum and the rounding of each exactly): 2a and 2c are small integers,
   * nibble r of 0x02242 is 2a (2 4 2 2 0 for r = 0..4), of 0x23220 is 2c (0 2 2 3 2).  Both products as
   * one packed multiply, both sums as one packed add: (2a q, 2c q) + (-2c, 2a). */
#if defined(__HIP_DEVICE_COMPILE__)
  const float a2 = (float)__builtin_amdgcn_ubfe(0x02242u, 4u * (unsigned)r, 4u);
  const float c2 = (float)__builtin_amdgcn_ubfe(0x23220u, 4u * (unsigned)r, 4u);
#else
  const float a2 = (float)((0x02242u >> (4u * (unsigned)r)) & 0xfu);
  const float c2 = (float)((0x23220u >> (4u * (unsigned)r)) & 0xfu);
#endif
  const fmd_v2f ac = {a2, c2};
  const fmd_v2f nd = fmd_pk_add_cross(ac * qc, ac);
  const float num = nd[0];
  const float den = nd[1]; /* 2, 4+2q, 2q+2, 2+3q or 2q: within [0.875, 2^26] */
  const float xr = fmd_div_midrange(num, den);
  const float z = xr * xr;
  const float w = z * z;
  /* s1 = z (aT0 + w (aT2 + w (aT4 + w (aT6 + w (aT8 + w aT10))))), s2 = w (aT1 + w (aT3 + w (aT5 +
   * w (aT7 + w aT9)))) as ONE packed Horner chain (s2's, s1's): s2's chain starts a step later, i.e.
   * with a leading 0 (0 w + aT9 = aT9 exactly).  On the device one block of eleven packed
   * instructions with the running pair as the FIRST factor of every multiply: gfx950 needs a wait
   * slot in front of a packed multiply whose first operand is a broadcast half and whose second is the
   * previous instruction's result -- the order the compiler picks for this chain (five s_nop) --
   * and none in this order (what it emits itself for `pair * scalar`). */
  const fmd_v2f wz = {w, z};
  fmd_v2f hc;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, %3\n\t"
      "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %5\n\t"
      "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %6\n\t"
      "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %7\n\t"
      "v_pk_mul_f32 %0, %0, %1"
      : "=&v"(hc)
      : "v"(wz), "s"(fmd_pack2f(0.0f, aT10)), "s"(fmd_pack2f(aT9, aT8)), "s"(fmd_pack2f(aT7, aT6)),
        "s"(fmd_pack2f(aT5, aT4)), "s"(fmd_pack2f(aT3, aT2)), "s"(fmd_pack2f(aT1, aT0)));
#else
  hc = (fmd_v2f){0.0f, aT10} * w + (fmd_v2f){aT9, aT8};
  hc = hc * w + (fmd_v2f){aT7, aT6};
  hc = hc * w + (fmd_v2f){aT5, aT4};
  hc = hc * w + (fmd_v2f){aT3, aT2};
  hc = hc * w + (fmd_v2f){aT1, aT0};
  hc = hc * wz; /* (s2, s1) */
#endif
  const float p = xr * (hc[1] + hc[0]);
  const float at = t[4] - ((p - t[5]) - xr); /* atanf(|y/x|) */
  /* quadrant: x >= 0 -> at, x < 0 -> pi - (at - pi_lo); then the sign of y (m = 1, 3 negate) */
  const float left = pi - (at - pi_lo);
  const float base = ((int32_t)fmd_f2u(x) < 0) ? left : at;
  *rare_measure = rare_m;
  return fmd_u2f(fmd_f2u(base) ^ (fmd_f2u(y) & 0x80000000u));
}

FMD_HD float fmd_atan2f_tab(float y, float x, const float* tab)
{
  uint32_t rare_m;
  float res = fmd_atan2f_tab_core(y, x, tab, &rare_m);
  const bool rare = rare_m >= FMD_ATAN_RARE_LIMIT;
  if (FMD_ANY_LANE(rare))
  {
    if (rare)
      res = fmd_atan2f(y, x);
  }
  return res;
}

/* Table-driven variant of fmd_sincos_nco for the per-sample loops: 1024-entry table of
 * (sin, cos)(k*2pi/1024) in double (built on the host in long double), argument split
 * phase = k*h + r with |r| <= h/2 = 0.0031 by a two-term Cody-Waite step, then
 * sin(kh+r) = S + (S*cm1 + C*sr), cos(kh+r) = C + (C*cm1 - S*sr) with the short series
 * sr = r - r^3/6 + r^5/120, cm1 = -r^2/2 + r^4/24 (truncation < 2e-18).  21 FP64 ops, no
 * quadrant logic; result rounded once to float like fmd_sincos_nco. */
struct FmdSincosTab
{
  double inv_h, h_hi, h_lo;
};
#define FMD_SINCOS_TAB_SIZE 1024
FMD_HD void fmd_sincos_tab(float phase, const double* tab /* [1024][2] */, struct FmdSincosTab t,
                           float* s, float* c)
{
  const double x = (double)phase;
  /* k = x / h rounded to an integer, as the low word of x / h + 1.5 * 2^52 (one operation instead
   * of multiply, round, convert; |x / h| < 2^31) */
  const double magic = 6755399441055744.0;
  const double tt = __builtin_fma(x, t.inv_h, magic);
  const double kf = tt - magic;
  uint64_t ttu;
  memcpy(&ttu, &tt, 8);
  const int k = (int)(uint32_t)ttu;
  double r = __builtin_fma(-kf, t.h_hi, x);
  r = __builtin_fma(-kf, t.h_lo, r);
  const double S = tab[2 * (k & (FMD_SINCOS_TAB_SIZE - 1))];
  const double C = tab[2 * (k & (FMD_SINCOS_TAB_SIZE - 1)) + 1];
  const double r2 = r * r;
  const double sr = __builtin_fma(r * r2, __builtin_fma(r2, 1.0 / 120.0, -1.0 / 6.0), r);
  const double cm1 = r2 * __builtin_fma(r2, 1.0 / 24.0, -0.5);
  const double so = S + __builtin_fma(C, sr, S * cm1);
  const double co = C + __builtin_fma(-S, sr, C * cm1);
  *s = (float)so;
  *c = (float)co;
}

/* The same for a phase known to lie in [0, 8): the table is (sin, cos)(k / 256), k = 0 .. 2047, and the
 * split phase = k / 256 + r is done in FLOAT without any rounding error: phase + 1.5 * 2^15 has an ulp
 * of 2^-8, i.e. the sum IS phase rounded to a multiple of 1 / 256 with k in its low mantissa bits;
 * taking the constant off again and subtracting from phase are both exact (|r| <= 2^-9).  Three float
 * operations and one conversion instead of a conversion and five double operations, and r carries no
 * error at all.  The two NCOs of the serial stage keep their phase in [0, 2 pi] (FmDecode.cpp:404-407,
 * :215-216).  NaN / infinite phases give NaN like fsincos; the table index is masked. */
/* x * a + b with two constants that are no inline literals: as ONE instruction, a in a scalar
 * register pair and b in a vector one (the compiler's choice is a register copy plus a two-operand
 * multiply-add; in loops whose cost is their instruction count that is one too many) */
FMD_HD double fmd_fma_const(double x, double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(a), "v"(b));
  return r;
#else
  return __builtin_fma(x, a, b);
#endif
}

#define FMD_SINCOS_P256_SIZE 2048
/* The two halves of the evaluation, for loops that issue the table read of the NEXT phase as soon as
 * that phase is known and finish it an iteration later (the read's latency, ~70 cycles for a lone wave,
 * then lies under independent work instead of at the head of the sample). */
struct FmdSincosP256
{
  double S, C; /* (sin, cos)(k / 256) */
  float r;     /* phase - k / 256, exact */
};
FMD_HD struct FmdSincosP256 fmd_sincos_p256_lookup(float phase, const double* tab /* [2048][2] */)
{
  const float big = 49152.0f;
  const float t = phase + big;
  const uint32_t k = fmd_f2u(t) & (FMD_SINCOS_P256_SIZE - 1);
  const float kx = t - big;
  struct FmdSincosP256 e;
  e.r = phase - kx;
  e.S = tab[2 * k];
  e.C = tab[2 * k + 1];
  return e;
}
/* The same lookup for a table in LDS: with 2^15 instead of 1.5 * 2^15 as the rounding constant (the
 * phases here are never negative) the low 24 bits of the sum ARE k, so the entry's byte offset is one
 * 24-bit multiply instead of a shift and a mask.  A NaN or infinite phase makes an offset beyond the
 * LDS allocation: such a read returns 0 and the result is NaN through r anyway. */
FMD_HD struct FmdSincosP256 fmd_sincos_p256_lookup_lds(float phase, const double* tab)
{
  const float big = 32768.0f;
  const float t = phase + big;
  uint32_t off; /* (the compiler turns a multiply by 16 back into shift + mask) */
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mul_u32_u24 %0, %1, 16" : "=v"(off) : "v"(t));
#else
  off = (fmd_f2u(t) & 0xffffffu) * 16u;
#endif
  const float kx = t - big;
  struct FmdSincosP256 e;
  e.r = phase - kx;
  typedef double fmd_v2d_ __attribute__((vector_size(16))); /* 16-byte aligned: one 16-byte read */
  const fmd_v2d_ sc = *(const fmd_v2d_*)((const char*)tab + off);
  e.S = sc[0];
  e.C = sc[1];
  return e;
}
/* m16 = -1 / 6: passed in so that a loop can keep it in a vector register (see fmd_fma_const) */
FMD_HD void fmd_sincos_p256_finish(struct FmdSincosP256 e, double m16, float* s, float* c)
{
  const double r = (double)e.r, S = e.S, C = e.C;
  const double r2 = r * r;
#if defined(__HIP_DEVICE_COMPILE__)
  /* r + r^3 (r^2 / 120 - 1 / 6): both multiply-adds in one block, so that the wait slot the compiler
   * puts behind an inline-asm result falls on an instruction that does not need it (the next one
   * here is the cosine series, independent of this) */
  double sr, st;
  asm("v_fma_f64 %1, %2, %3, %4\n\t"
      "v_fma_f64 %0, %5, %1, %6"
      : "=&v"(sr), "=&v"(st)
      : "v"(r2), "s"(1.0 / 120.0), "v"(m16), "v"(r * r2), "v"(r));
#else
  const double sr = __builtin_fma(r * r2, fmd_fma_const(r2, 1.0 / 120.0, m16), r);
#endif
  const double cm1 = r2 * __builtin_fma(r2, 1.0 / 24.0, -0.5);
  const double so = S + __builtin_fma(C, sr, S * cm1);
  const double co = C + __builtin_fma(-S, sr, C * cm1);
  *s = (float)so;
  *c = (float)co;
}
FMD_HD void fmd_sincos_p256k(float phase, const double* tab /* [2048][2] */, double m16, float* s, float* c)
{
  fmd_sincos_p256_finish(fmd_sincos_p256_lookup(phase, tab), m16, s, c);
}
FMD_HD void fmd_sincos_p256(float phase, const double* tab /* [2048][2] */, float* s, float* c)
{
  fmd_sincos_p256k(phase, tab, -1.0 / 6.0, s, c);
}

/* RDSProcess.cpp:187-217 */
FMD_HD float fmd_rds_arctan2(float y, float x)
{
  if (x == 0.0f)
  {
    if (y > 0.0f)
      return (float)FMD_K_PI2;
    if (y == 0.0f)
      return 0.0f;
    return (float)-FMD_K_PI2;
  }
  float angle;
  float z = y / x;
  if (fabsf(z) < 1.0f)
  {
    angle = (float)((double)z / (1.0 + 0.2854 * (double)z * (double)z));
    if (x < 0.0f)
    {
      if (y < 0.0f)
        return (float)((double)angle - FMD_K_PI);
      return (float)((double)angle + FMD_K_PI);
    }
  }
  else
  {
    angle = (float)(FMD_K_PI2 - (double)z / ((double)(z * z) + 0.2854));
    if (y < 0.0f)
      return (float)((double)angle - FMD_K_PI);
  }
  return angle;
}

#endif
