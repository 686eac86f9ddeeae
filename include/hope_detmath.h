/*
 * hope_detmath.h — deterministic subset of libm used where the reference calls
 * atan2f/cosf/sinf (pcl::computeRoots inside pcl::eigen33, reached from
 * Utilities::estimateNorm, utilities.cpp:428-446) and asinf/acosf
 * (Utilities::normalAnalysis, utilities.cpp:171,231).
 *
 * glibc's and ROCm OCML's float trig differ in the last bit, and the results feed
 * discontinuous gates (|nz| > th_norm, grad > th_angle), so the numerical contract of
 * the C ABI is defined on these functions instead: evaluated in IEEE double with only
 * + - * / sqrt (no FMA contraction: compile with -ffp-contract=off), then rounded once
 * to float.  Every constant is a small exactly-representable integer or a literal both
 * compilers round identically.  Host (g++) and device (hipcc, gfx950) therefore produce
 * identical bits; tests/test_detmath.py checks that they are the correctly rounded float
 * values on dense samples and never more than 1 float ulp from glibc (whose own float
 * trig is off by one ulp in 1-16 % of cases).
 */
#ifndef HOPE_DETMATH_H
#define HOPE_DETMATH_H

#if defined(__HIPCC__)
#define HOPE_HD __host__ __device__ inline
#else
#define HOPE_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define HOPE_SQRT_F64(x) __builtin_sqrt(x)
#else
#include <math.h>
#define HOPE_SQRT_F64(x) sqrt(x)
#endif

#define HOPE_PI_F64 3.14159265358979323846
#define HOPE_PI_2_F64 1.57079632679489661923

/* atan(t) for t in [0,1]: two half-angle steps atan(t) = 2 atan(t / (1 + sqrt(1 + t^2)))
 * bring the argument below tan(pi/16) = 0.199, then a 15-term Maclaurin series. */
HOPE_HD double hope_det_atan_unit(double t)
{
  double s = HOPE_SQRT_F64(1.0 + t * t);
  t = t / (1.0 + s);
  s = HOPE_SQRT_F64(1.0 + t * t);
  t = t / (1.0 + s);
  const double z = t * t;
  double p = 1.0 / 29.0;
  p = 1.0 / 27.0 - z * p;
  p = 1.0 / 25.0 - z * p;
  p = 1.0 / 23.0 - z * p;
  p = 1.0 / 21.0 - z * p;
  p = 1.0 / 19.0 - z * p;
  p = 1.0 / 17.0 - z * p;
  p = 1.0 / 15.0 - z * p;
  p = 1.0 / 13.0 - z * p;
  p = 1.0 / 11.0 - z * p;
  p = 1.0 / 9.0 - z * p;
  p = 1.0 / 7.0 - z * p;
  p = 1.0 / 5.0 - z * p;
  p = 1.0 / 3.0 - z * p;
  p = 1.0 - z * p;
  return 4.0 * (t * p);
}

/* atan2 for finite arguments (NaN in -> NaN out). */
HOPE_HD double hope_det_atan2(double y, double x)
{
  if (!(y == y) || !(x == x)) return y + x;
  const double ax = x < 0.0 ? -x : x;
  const double ay = y < 0.0 ? -y : y;
  double a;
  if (ax == 0.0 && ay == 0.0) {
    a = 0.0;
  } else if (ay > ax) {
    a = HOPE_PI_2_F64 - hope_det_atan_unit(ax / ay);
  } else {
    a = hope_det_atan_unit(ay / ax);
  }
  if (x < 0.0) a = HOPE_PI_F64 - a;
  return y < 0.0 ? -a : a;
}

/* cos / sin for |x| <= pi/2 (computeRoots only needs [0, pi/3]); nested Maclaurin
 * series.  The coefficients 1 / ((2k-1)(2k)) and 1 / ((2k)(2k+1)) are compile-time constants (the
 * correctly rounded doubles, the same on host and device): a multiplication each, no division --
 * 26 double-precision divisions per normal were a third of k_normals_tile's instructions. */
HOPE_HD double hope_det_cos(double x)
{
  const double z = x * x;
  double p = 1.0;
  p = 1.0 - z * (1.0 / 650.0) * p; /* 25*26 */
  p = 1.0 - z * (1.0 / 552.0) * p; /* 23*24 */
  p = 1.0 - z * (1.0 / 462.0) * p;
  p = 1.0 - z * (1.0 / 380.0) * p;
  p = 1.0 - z * (1.0 / 306.0) * p;
  p = 1.0 - z * (1.0 / 240.0) * p;
  p = 1.0 - z * (1.0 / 182.0) * p;
  p = 1.0 - z * (1.0 / 132.0) * p;
  p = 1.0 - z * (1.0 / 90.0) * p;
  p = 1.0 - z * (1.0 / 56.0) * p;
  p = 1.0 - z * (1.0 / 30.0) * p;
  p = 1.0 - z * (1.0 / 12.0) * p;
  p = 1.0 - z * (1.0 / 2.0) * p;
  return p;
}

HOPE_HD double hope_det_sin(double x)
{
  const double z = x * x;
  double p = 1.0;
  p = 1.0 - z * (1.0 / 702.0) * p; /* 26*27 */
  p = 1.0 - z * (1.0 / 600.0) * p; /* 24*25 */
  p = 1.0 - z * (1.0 / 506.0) * p;
  p = 1.0 - z * (1.0 / 420.0) * p;
  p = 1.0 - z * (1.0 / 342.0) * p;
  p = 1.0 - z * (1.0 / 272.0) * p;
  p = 1.0 - z * (1.0 / 210.0) * p;
  p = 1.0 - z * (1.0 / 156.0) * p;
  p = 1.0 - z * (1.0 / 110.0) * p;
  p = 1.0 - z * (1.0 / 72.0) * p;
  p = 1.0 - z * (1.0 / 42.0) * p;
  p = 1.0 - z * (1.0 / 20.0) * p;
  p = 1.0 - z * (1.0 / 6.0) * p;
  return x * p;
}

/* float front-ends: the reference's float libm calls */
HOPE_HD float hope_det_atan2f(float y, float x) { return (float)hope_det_atan2((double)y, (double)x); }
HOPE_HD float hope_det_cosf(float x) { return (float)hope_det_cos((double)x); }
HOPE_HD float hope_det_sinf(float x) { return (float)hope_det_sin((double)x); }

/* asinf / acosf; |x| > 1 gives NaN like libm */
HOPE_HD float hope_det_asinf(float xf)
{
  const double x = (double)xf;
  if (!(x >= -1.0 && x <= 1.0)) return (float)((x - x) / (x - x));
  const double c = HOPE_SQRT_F64((1.0 - x) * (1.0 + x));
  return (float)hope_det_atan2(x, c);
}

HOPE_HD float hope_det_acosf(float xf)
{
  const double x = (double)xf;
  if (!(x >= -1.0 && x <= 1.0)) return (float)((x - x) / (x - x));
  const double c = HOPE_SQRT_F64((1.0 - x) * (1.0 + x));
  return (float)hope_det_atan2(c, x);
}

#endif /* HOPE_DETMATH_H */
