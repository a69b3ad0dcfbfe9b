"""csrc/dual.cuh (forward-mode differentiation of user-defined log-densities) compiled as plain C++ on the CPU: every
overloaded operator / function against central differences, and a template density instantiated with double and with
Dual (the way the run-time compiled kernels use it).  Reference: aehmc/hmc.py:33-34 (aesara.grad of the potential)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cmath>
#include "dual.cuh"
using aehmc::Dual;
template <class T> T density(T q, double a) {   // a user-style template body: mixes T and double, uses ADL for the functions
  T z = (q - a) / 1.5;
  T lp = -0.5 * z * z - log1p(exp(-q)) + sqrt(1.0 + square(z)) * 0.25 - softplus(z) + 0.1 * tanh(q) - erf(z) * 0.2;
  lp += pow(1.0 + q * q, 0.3) - log(2.0 + cos(q)) + sin(z) / (2.0 + z * z) - expm1(-z * z) + fabs(q - 0.2);
  lp -= pow(1.0 + z * z, z * 0.1);
  if (q > 10.0) lp = lp * 2.0;
  return lp;
}
int main() {
  int bad = 0;
  for (double q = -2.0; q <= 2.0; q += 0.37) {
    const double h = 1e-6;
    const Dual r = density(Dual(q, 1.0), 0.3);
    const double v = density(q, 0.3), fd = (density(q + h, 0.3) - density(q - h, 0.3)) / (2 * h);
    if (std::fabs(r.v - v) > 1e-14 * (1 + std::fabs(v)) || std::fabs(r.d - fd) > 2e-8 * (1 + std::fabs(fd))) {
      std::printf("q=%g value %g / %g derivative %.12g / %.12g\n", q, r.v, v, r.d, fd);
      bad++;
    }
  }
  // quotient, compound assignment, comparisons
  Dual a(2.0, 1.0), b(3.0, 0.0);
  Dual c = a / b; c += a; c *= 2.0; c -= 1.0; c /= b;     // ((a/3 + a) * 2 - 1) / 3 -> d/da = (1/3 + 1) * 2 / 3
  if (std::fabs(c.d - (1.0 / 3 + 1) * 2 / 3) > 1e-15 || !(a < b) || !(b >= a) || a == b || !(a != 3.0) || !(1.0 < a)) bad++;
  Dual d = 5.0 / a;                                        // -5 / a^2
  if (std::fabs(d.d + 5.0 / 4.0) > 1e-15 || std::fabs(aehmc::ad::value_of(d) - 2.5) > 1e-15) bad++;
  // lgamma / digamma (round 6): derivative against central differences, digamma against known values
  for (double x = 0.3; x < 40.0; x *= 1.7) {
    const Dual r = lgamma(Dual(x, 1.0));
    const double h = 1e-6 * x, fd = (std::lgamma(x + h) - std::lgamma(x - h)) / (2 * h);
    if (std::fabs(r.v - std::lgamma(x)) > 2e-14 * (1 + std::fabs(r.v)) || std::fabs(r.d - fd) > 3e-8 * (1 + std::fabs(fd))) { std::printf("lgamma x=%g %.12g / %.12g\n", x, r.d, fd); bad++; }
  }
  if (std::fabs(aehmc::ad::digamma(1.0) + 0.57721566490153286) > 2e-15 || std::fabs(aehmc::ad::digamma(0.5) + 1.9635100260214235) > 4e-15 ||
      std::fabs(aehmc::ad::digamma(100.0) - 4.6001618527380874) > 2e-15 || std::fabs(aehmc::ad::digamma(-0.5) - 0.03648997397857652) > 4e-15) bad++;
  std::printf("bad=%d\n", bad);
  return bad;
}
"""


def test_dual_numbers_against_finite_differences(tmp_path):
    src = tmp_path / "dual_test.cpp"
    src.write_text(SRC)
    exe = tmp_path / "dual_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout


def test_digamma_against_scipy(tmp_path):
    import numpy as np
    from scipy.special import digamma
    xs = np.concatenate([np.geomspace(1e-3, 1e3, 200), -np.linspace(0.1, 5.9, 30) - 0.05])
    src = tmp_path / "dg.cpp"
    src.write_text('#include <cstdio>\n#include <cmath>\n#include "dual.cuh"\nint main() { double x; while (std::scanf("%lf", &x) == 1) '
                   'std::printf("%.17g\\n", aehmc::ad::digamma(x)); return 0; }\n')
    exe = tmp_path / "dg"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], input="\n".join(repr(float(x)) for x in xs), capture_output=True, text=True, check=True).stdout.split()
    got = np.array([float(v) for v in out])
    np.testing.assert_allclose(got, digamma(xs), rtol=5e-14, atol=5e-15)


def test_softplus_and_logistic_against_numpy(tmp_path):
    """dual.cuh's softplus (one exponential + the 17-term atanh series for log(1 + e), e <= 1) and logistic function:
    within 4 ulp of numpy.logaddexp(0, x) / scipy.special.expit over the whole range, exact limits at +-inf"""
    import numpy as np
    from scipy.special import expit
    xs = np.concatenate([np.linspace(-745.0, 745.0, 1501), np.linspace(-40.0, 40.0, 4001), np.random.default_rng(0).normal(0, 3, 4000),
                         [0.0, -0.0, 1e-300, -1e-300, np.inf, -np.inf]])
    src = tmp_path / "sp.cpp"
    src.write_text('#include <cstdio>\n#include <cmath>\n#include "dual.cuh"\nint main() { double x; while (std::scanf("%lf", &x) == 1) { '
                   'const aehmc::Dual r = softplus(aehmc::Dual(x, 1.0)); const aehmc::Dual l = logistic(aehmc::Dual(x, 1.0)); '
                   'std::printf("%.17g %.17g %.17g %.17g %.17g\\n", aehmc::ad::softplus(x), r.v, r.d, l.v, l.d); } return 0; }\n')
    exe = tmp_path / "sp"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], input="\n".join(repr(float(x)) for x in xs), capture_output=True, text=True, check=True).stdout.split()
    got = np.array([float(v) for v in out]).reshape(len(xs), 5)
    want = np.logaddexp(0.0, xs)
    assert np.array_equal(got[:, 0], got[:, 1])
    np.testing.assert_array_max_ulp(got[:, 0], want, maxulp=4)
    with np.errstate(over="ignore"):
        s = expit(xs)
    np.testing.assert_allclose(got[:, 2], s, rtol=2e-15, atol=1e-300)  # (expit flushes the subnormal tail to 0)
    np.testing.assert_allclose(got[:, 3], s, rtol=2e-15, atol=1e-300)
    np.testing.assert_allclose(got[:, 4], s * (1.0 - s), rtol=4e-15, atol=1e-300)
    assert got[-2, 0] == np.inf and got[-1, 0] == 0.0 and got[-2, 2] == 1.0 and got[-1, 2] == 0.0


def test_fast_log_and_log1p_against_numpy(tmp_path):
    """dual.cuh's log_fast / log1p_fast (what log / log1p of a user density compile to): within 2 ulp of numpy over the whole
    range, the library's results at the edges (0, negative, subnormal, inf, nan)"""
    import numpy as np
    r = np.random.default_rng(1)
    xs = np.concatenate([np.exp(r.uniform(-700, 700, 20000)), 1.0 + r.normal(0, 1e-3, 5000), r.uniform(0.5, 2.0, 20000), np.exp2(np.arange(-1022, 1024)),
                         np.nextafter(np.sqrt(2.0) * np.exp2(np.arange(-5, 6)), 0), np.nextafter(np.sqrt(2.0) * np.exp2(np.arange(-5, 6)), 9),
                         [0.0, -1.0, 5e-324, 1e-310, np.inf, np.nan, 1.0, np.nextafter(1.0, 0), np.nextafter(1.0, 2)]])
    ys = np.concatenate([np.expm1(r.uniform(-36, 700, 20000)), r.normal(0, 1e-3, 5000), r.uniform(-0.999999, 3.0, 20000), -np.exp(r.uniform(-700, 0, 5000)),
                         np.exp(r.uniform(-745, -30, 5000)), [0.0, -0.0, -1.0, -2.0, 1e-17, -1e-17, 2.0 ** -53, -2.0 ** -54, np.inf, np.nan, 1.0, 3.0,
                                                              np.nextafter(-1.0, 0), 1.7976931348623157e308]])
    src = tmp_path / "lg.cpp"
    src.write_text('#include <cstdio>\n#include <cmath>\n#include "dual.cuh"\nint main() { int n; double x; std::scanf("%d", &n); '
                   'for (int i = 0; i < n; i++) { std::scanf("%lf", &x); std::printf("%.17g\\n", aehmc::ad::log_fast(x)); } '
                   'while (std::scanf("%lf", &x) == 1) std::printf("%.17g\\n", aehmc::ad::log1p_fast(x)); return 0; }\n')
    exe = tmp_path / "lg"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    inp = f"{len(xs)}\n" + "\n".join(repr(float(x)) for x in np.concatenate([xs, ys]))
    out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.split()
    got = np.array([float(v) for v in out])
    with np.errstate(all="ignore"):
        wl, wp = np.log(xs), np.log1p(ys)
    gl, gp = got[:len(xs)], got[len(xs):]
    for g, w in ((gl, wl), (gp, wp)):
        fin = np.isfinite(w)
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(g[np.isinf(w)], w[np.isinf(w)])
        np.testing.assert_array_max_ulp(g[fin], w[fin], maxulp=2)


def test_fast_lgamma_against_scipy(tmp_path):
    """dual.cuh's lgamma_fast (shift to x >= 10 + Stirling's series): absolute error below 1e-14 + 4 ulp against
    scipy.special.gammaln on (1e-300, 1e300), the library's result outside"""
    import numpy as np
    from scipy.special import gammaln
    r = np.random.default_rng(2)
    xs = np.concatenate([np.geomspace(1e-6, 1e6, 4000), r.uniform(0.0, 30.0, 20000), np.exp(r.uniform(-690, 690, 5000)), np.arange(1, 60) * 0.5,
                         [1.0, 2.0, np.nextafter(1.0, 2), np.nextafter(2.0, 1), 10.0, np.nextafter(10.0, 0), 1e-300, 1e300, 0.0, -0.5, -2.5, np.inf, np.nan]])
    src = tmp_path / "lgm.cpp"
    src.write_text('#include <cstdio>\n#include <cmath>\n#include "dual.cuh"\nint main() { double x; while (std::scanf("%lf", &x) == 1) '
                   'std::printf("%.17g\\n", aehmc::ad::lgamma_fast(x)); return 0; }\n')
    exe = tmp_path / "lgm"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], input="\n".join(repr(float(x)) for x in xs), capture_output=True, text=True, check=True).stdout.split()
    got = np.array([float(v) for v in out])
    with np.errstate(all="ignore"):
        want = gammaln(xs)
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])
    err = np.abs(got[fin] - want[fin])
    assert np.all(err <= 1e-14 + 4 * np.spacing(np.abs(want[fin]))), (xs[fin][np.argmax(err)], err.max())
