// Forward-mode differentiation of user-defined log-densities (gfx950; also compiles as plain C++ for the CPU tests).
//
// The reference takes any `logprob_fn` and differentiates it symbolically (aehmc/hmc.py:33-34: aesara.grad of the
// potential; integrators.py:61-65).  Here the user writes the log-density ONCE, as a function template over its
// arithmetic type, and the engine instantiates it with `aehmc::Dual` (value + one directional derivative) inside the
// run-time compiled kernels:
//   * coordinate-wise targets (targets.Custom):    template <class T> T aehmc_logp(T q, long long i, const double *const *prm)
//     -- coordinate i's term; one Dual evaluation per element gives u_i = -logp_i and du_i/dq_i;
//   * row-reduction targets (targets.CustomGLM):   template <class T> T aehmc_glm_loglik(T z, double y, long long n, prm)
//                                                  template <class T> T aehmc_glm_logprior(T q, long long i, prm);
//   * joint (non-separable) targets, D <= 64 (targets.CustomJoint):
//                                                  template <class V> auto aehmc_logp(const V &q, const double *const *prm)
//     with q[i] the coordinates and q.size() their number: lane i of the chain's wavefront holds q_i and evaluates the
//     density with the derivative seeded at ITS coordinate -- the D forward passes of the gradient run side by side
//     in the lanes, every lane ends with the same value and with dlogp/dq_lane (JointArg below).
// Supported: + - * / (Dual with Dual or double), unary -, comparisons (on the value), exp, log, log1p, expm1, sqrt,
// pow(x, double), pow(x, y), sin, cos, tanh, fabs, erf, lgamma (with digamma below), square(x), softplus(x).
#pragma once
#if defined(__HIPCC_RTC__) || defined(__HIPCC__)
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime and the math functions itself) */
#include <hip/hip_runtime.h>
#endif
#define AEHMC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#include <cmath>
#define AEHMC_HD inline
#endif

namespace aehmc {
namespace ad {  // (its own namespace: found by argument-dependent lookup from user code, invisible to the engine's own exp / log1p calls)

struct Dual {
  double v, d;
  AEHMC_HD Dual() : v(0.0), d(0.0) {}
  AEHMC_HD Dual(double value) : v(value), d(0.0) {}  // a constant
  AEHMC_HD Dual(double value, double deriv) : v(value), d(deriv) {}
};

AEHMC_HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
AEHMC_HD Dual operator+(Dual a) { return a; }
AEHMC_HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
AEHMC_HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
AEHMC_HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
AEHMC_HD Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return Dual(q, (a.d - q * b.d) / b.v);
}
AEHMC_HD Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
AEHMC_HD Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
AEHMC_HD Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
AEHMC_HD Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
AEHMC_HD Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
AEHMC_HD Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
AEHMC_HD Dual operator/(Dual a, double b) { return Dual(a.v / b, a.d / b); }
AEHMC_HD Dual operator/(double a, Dual b) {
  const double q = a / b.v;
  return Dual(q, -q * b.d / b.v);
}
AEHMC_HD Dual &operator+=(Dual &a, Dual b) { a = a + b; return a; }
AEHMC_HD Dual &operator-=(Dual &a, Dual b) { a = a - b; return a; }
AEHMC_HD Dual &operator*=(Dual &a, Dual b) { a = a * b; return a; }
AEHMC_HD Dual &operator/=(Dual &a, Dual b) { a = a / b; return a; }
AEHMC_HD Dual &operator+=(Dual &a, double b) { a.v += b; return a; }
AEHMC_HD Dual &operator-=(Dual &a, double b) { a.v -= b; return a; }
AEHMC_HD Dual &operator*=(Dual &a, double b) { a = a * b; return a; }
AEHMC_HD Dual &operator/=(Dual &a, double b) { a = a / b; return a; }
#define AEHMC_DUAL_CMP(op)                                            \
  AEHMC_HD bool operator op(Dual a, Dual b) { return a.v op b.v; }    \
  AEHMC_HD bool operator op(Dual a, double b) { return a.v op b; }    \
  AEHMC_HD bool operator op(double a, Dual b) { return a op b.v; }
AEHMC_DUAL_CMP(<)
AEHMC_DUAL_CMP(>)
AEHMC_DUAL_CMP(<=)
AEHMC_DUAL_CMP(>=)
AEHMC_DUAL_CMP(==)
AEHMC_DUAL_CMP(!=)
#undef AEHMC_DUAL_CMP

// elementary functions; the double overloads let one template body serve both instantiations
AEHMC_HD double square(double x) { return x * x; }
AEHMC_HD Dual square(Dual x) { return Dual(x.v * x.v, 2.0 * x.v * x.d); }
AEHMC_HD Dual exp(Dual x) {
  const double e = ::exp(x.v);
  return Dual(e, e * x.d);
}
AEHMC_HD Dual expm1(Dual x) { return Dual(::expm1(x.v), ::exp(x.v) * x.d); }
// log and log1p of user densities (round 6).  The device library's fp64 log / log1p are 98 / 135 vector instructions
// (double-double arithmetic for a correctly rounded result); a density that calls one per data row or coordinate -- Student-t,
// Cauchy, log-normal, gamma -- is bound by it.  These take ~45 / ~60: x = 2^k m with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m - 1) / (m + 1) (|s| <= 0.1716: ten terms of the odd series, truncation below 1e-18), k ln 2 added
// in two pieces; log1p(x) = log(u) + (x - (u - 1)) / u with u = fl(1 + x) (the rounding of the sum put back).  Within 2 ulp
// of numpy over the whole range (tests/test_dual.py); zero, negative, subnormal, infinite and NaN arguments go to the
// library.  The engine's own arithmetic (tree weights, acceptance probabilities: nuts_tree.cuh) does not use these: it
// reproduces numpy's bits.
AEHMC_HD double log_fast(double x) {
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return ::log(x);
  long long b;
  __builtin_memcpy(&b, &x, 8);
  int k = (int)(b >> 52) - 1023;
  b = (b & 0x000fffffffffffffLL) | 0x3ff0000000000000LL;
  double m;
  __builtin_memcpy(&m, &b, 8);
  if (m > 1.4142135623730951) {
    m *= 0.5;
    k += 1;
  }
  const double s = (m - 1.0) / (m + 1.0), w = s * s;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, w, 1.0 / 19.0);
  p = __builtin_fma(p, w, 1.0 / 17.0);
  p = __builtin_fma(p, w, 1.0 / 15.0);
  p = __builtin_fma(p, w, 1.0 / 13.0);
  p = __builtin_fma(p, w, 1.0 / 11.0);
  p = __builtin_fma(p, w, 1.0 / 9.0);
  p = __builtin_fma(p, w, 1.0 / 7.0);
  p = __builtin_fma(p, w, 1.0 / 5.0);
  p = __builtin_fma(p, w, 1.0 / 3.0);
  const double s2 = s + s, r = __builtin_fma(s2 * w, p, s2), kd = (double)k;
  return __builtin_fma(kd, 6.93147180369123816490e-01, __builtin_fma(kd, 1.90821492927058770002e-10, r));
}
AEHMC_HD double log1p_fast(double x) {
  const double u = 1.0 + x;
  if (!(u >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return ::log1p(x);
  if (u == 1.0) return x;  // |x| < 2^-53: log1p(x) = x to the last bit
  const double c = (u >= 2.0 ? 1.0 - (u - x) : x - (u - 1.0)) / u;
  return log_fast(u) + c;
}
AEHMC_HD Dual log(Dual x) { return Dual(log_fast(x.v), x.d / x.v); }
AEHMC_HD Dual log1p(Dual x) { return Dual(log1p_fast(x.v), x.d / (1.0 + x.v)); }
AEHMC_HD Dual sqrt(Dual x) {
  const double s = ::sqrt(x.v);
  return Dual(s, 0.5 * x.d / s);
}
AEHMC_HD Dual pow(Dual x, double p) {  // (one pow: the derivative p x^(p-1) = p f / x away from x = 0)
  const double f = ::pow(x.v, p);
  return Dual(f, (x.v != 0.0 ? p * f / x.v : p * ::pow(x.v, p - 1.0)) * x.d);
}
AEHMC_HD Dual pow(Dual x, Dual y) {
  const double f = ::pow(x.v, y.v);
  return Dual(f, f * (y.d * ::log(x.v) + y.v * x.d / x.v));
}
AEHMC_HD Dual sin(Dual x) { return Dual(::sin(x.v), ::cos(x.v) * x.d); }
AEHMC_HD Dual cos(Dual x) { return Dual(::cos(x.v), -::sin(x.v) * x.d); }
AEHMC_HD Dual tanh(Dual x) {
  const double t = ::tanh(x.v);
  return Dual(t, (1.0 - t * t) * x.d);
}
AEHMC_HD Dual sinh(Dual x) { return Dual(::sinh(x.v), ::cosh(x.v) * x.d); }
AEHMC_HD Dual cosh(Dual x) { return Dual(::cosh(x.v), ::sinh(x.v) * x.d); }
AEHMC_HD Dual atan(Dual x) { return Dual(::atan(x.v), x.d / (1.0 + x.v * x.v)); }
AEHMC_HD Dual fabs(Dual x) { return x.v < 0 ? -x : x; }
AEHMC_HD Dual erfc(Dual x) { return Dual(::erfc(x.v), -1.1283791670955126 * ::exp(-x.v * x.v) * x.d); }
AEHMC_HD Dual erf(Dual x) { return Dual(::erf(x.v), 1.1283791670955126 * ::exp(-x.v * x.v) * x.d); }
// digamma = d/dx lgamma(x) (round 6: Gamma / Beta / Student-t / negative-binomial densities with traced shape parameters):
// the recurrence psi(x) = psi(x + 1) - 1 / x up to x >= 10, there the asymptotic series ln x - 1 / (2x) - sum B_2k / (2k x^2k)
// to x^-14 (truncation error below 1e-17); for x <= 0 the reflection psi(x) = psi(1 - x) - pi / tan(pi x).
// Against scipy.special.digamma: 2e-15 relative on [1e-3, 1e3] (tests/test_dual.py).
AEHMC_HD double digamma(double x) {
  double refl = 0.0;
  if (x <= 0.0) {
    refl = -3.14159265358979323846 / ::tan(3.14159265358979323846 * x);
    x = 1.0 - x;
  }
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double i2 = 1.0 / (x * x);
  const double ser = i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0 -
                     i2 * (691.0 / 32760.0 - i2 * (1.0 / 12.0)))))));
  return refl + r + ::log(x) - 0.5 / x - ser;
}
// lgamma of a positive argument without the device library's 936 instructions (a negative-binomial or Student-t likelihood
// with a traced shape parameter calls it per data row): x is shifted up to x + n >= 10 (lgamma(x) = lgamma(x + n) -
// log(x (x + 1) ... (x + n - 1))), there Stirling's series to x^-15 (next term below 2e-18).  Absolute error below 1e-14 + 4 ulp
// (the shifted values around 15 are subtracted; tests/test_dual.py against scipy.special.gammaln) -- the accuracy a
// log-density needs, not a correctly rounded value near the zeros at 1 and 2.  x <= 0, tiny, infinite or NaN: the library.
AEHMC_HD double lgamma_fast(double x) {
  if (!(x >= 1e-300 && x <= 1e300)) return ::lgamma(x);
  double prod = 1.0;
  while (x < 10.0) {
    prod *= x;
    x += 1.0;
  }
  const double r = 1.0 / x, w = r * r;
  const double ser = r * (1.0 / 12.0 - w * (1.0 / 360.0 - w * (1.0 / 1260.0 - w * (1.0 / 1680.0 - w * (1.0 / 1188.0 -
                     w * (691.0 / 360360.0 - w * (1.0 / 156.0 - w * (3617.0 / 122400.0))))))));
  const double st = (x - 0.5) * log_fast(x) - x + 0.91893853320467274178 + ser;
  return prod == 1.0 ? st : st - log_fast(prod);
}
AEHMC_HD Dual lgamma(Dual x) { return Dual(lgamma_fast(x.v), digamma(x.v) * x.d); }
// log(1 + exp(x)) without overflow (the logistic log-likelihood's building block) and the logistic function.
// Value and derivative come from ONE exponential, e = exp(-|x|): softplus = max(x, 0) + log(1 + e), its derivative the
// logistic function 1 / (1 + e) or e / (1 + e) (round 6: the density-only logistic regression spent a third of its row
// function on a second exponential).  log(1 + e) for 0 <= e <= 1 is 2 atanh(s), s = e / (2 + e) <= 1/3: seventeen terms of
// the odd series (truncation below 2e-18 relative), about 30 instructions where the library's double-double log1p takes
// over a hundred -- the row function of a logistic regression went from 301 to 190 instructions (profiles/r6/INDEX.md).
// Within 4 ulp of numpy.logaddexp(0, x) (tests/test_dual.py).
#define AEHMC_ATANH_C(k) (1.0 / (double)(k))
AEHMC_HD double log1p_unit(double e) {
  const double s = e / (2.0 + e), w = s * s;
  double p = AEHMC_ATANH_C(33);
  p = __builtin_fma(p, w, AEHMC_ATANH_C(31));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(29));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(27));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(25));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(23));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(21));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(19));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(17));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(15));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(13));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(11));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(9));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(7));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(5));
  p = __builtin_fma(p, w, AEHMC_ATANH_C(3));
  const double s2 = s + s;
  return __builtin_fma(s2 * w, p, s2);
}
AEHMC_HD double softplus(double x) { return (x > 0 ? x : 0.0) + log1p_unit(::exp(x > 0 ? -x : x)); }
AEHMC_HD double logistic(double x) {
  const double e = ::exp(x > 0 ? -x : x);
  return (x > 0 ? 1.0 : e) / (1.0 + e);
}
AEHMC_HD Dual logistic(Dual x) {
  const double l = logistic(x.v);
  return Dual(l, x.d * (l * (1.0 - l)));
}
AEHMC_HD Dual softplus(Dual x) {
  const double e = ::exp(x.v > 0 ? -x.v : x.v);
  return Dual((x.v > 0 ? x.v : 0.0) + log1p_unit(e), x.d * ((x.v > 0 ? 1.0 : e) / (1.0 + e)));
}
AEHMC_HD double value_of(double x) { return x; }
AEHMC_HD double value_of(Dual x) { return x.v; }

}  // namespace ad
using ad::Dual;

#if defined(__HIPCC_RTC__) || defined(__HIPCC__)
// The coordinates of a chain as a joint density sees them: lane i of the wavefront holds q_i in `ql` (D <= 64).
// JointArg<Dual>::operator[](i) is q_i with derivative 1 in the lane that owns coordinate i (0 elsewhere): after one
// evaluation of the density every lane holds logp (the same bits) and dlogp/dq_lane.  Indices must be uniform over the
// wavefront -- they are in any density: its control flow depends on values, which are the same in every lane.
__device__ inline __attribute__((always_inline)) double joint_lane_read(double x, int i) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), i), __builtin_amdgcn_readlane(__double2loint(x), i));
}
template <class T>
struct JointArg;
template <>
struct JointArg<double> {
  double ql;
  int lane, D;
  __device__ inline __attribute__((always_inline)) double operator[](int i) const { return joint_lane_read(ql, i); }
  __device__ inline __attribute__((always_inline)) int size() const { return D; }
};
template <>
struct JointArg<Dual> {
  double ql;
  int lane, D;
  __device__ inline __attribute__((always_inline)) Dual operator[](int i) const { return Dual(joint_lane_read(ql, i), i == lane ? 1.0 : 0.0); }
  __device__ inline __attribute__((always_inline)) int size() const { return D; }
};
// The same view over a row of coordinates in memory (LDS) for joint densities of MORE than 64 coordinates (round 5;
// engine.cuh: k_target_joint_rows): the wavefront evaluates the density ceil(D / 64) times, lane l seeding coordinate
// `seed` = l + 64 k in pass k.  q[i] with a wave-uniform i is a broadcast read.
template <class T>
struct JointRow;
template <>
struct JointRow<double> {
  const double *q;
  int seed, D;
  __device__ inline __attribute__((always_inline)) double operator[](int i) const { return q[i]; }
  __device__ inline __attribute__((always_inline)) int size() const { return D; }
};
template <>
struct JointRow<Dual> {
  const double *q;
  int seed, D;
  __device__ inline __attribute__((always_inline)) Dual operator[](int i) const { return Dual(q[i], i == seed ? 1.0 : 0.0); }
  __device__ inline __attribute__((always_inline)) int size() const { return D; }
};
#endif

}  // namespace aehmc

// unqualified calls in user code (exp(x), log1p(x), ...) with x a Dual resolve by argument-dependent lookup; for T =
// double they are the device library's own functions
using aehmc::ad::square;
using aehmc::ad::softplus;
using aehmc::ad::logistic;
using aehmc::ad::value_of;
using aehmc::ad::digamma;
