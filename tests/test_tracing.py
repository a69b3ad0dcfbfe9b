"""aehmc_amd/tracing.py: a Python ``logprob_fn`` (reference: README.md:27-36, aehmc/hmc.py:16-40) traced into the
``aehmc_logp`` template.  The emitted source is compiled as plain C++ against csrc/dual.cuh (the way tests/test_dual.py
does) and evaluated with double and with Dual: value == the Python function on plain numpy arrays, gradient == central
differences of it.  (The GPU side -- the same sources through hipRTC, whole transitions against the numpy
restatement -- is tests/test_gpu_callable.py.)"""
import os
import subprocess

import numpy as np
import pytest

from aehmc_amd import targets, tracing

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = np.random.default_rng(5)
NU, SC = 3.0 + 5 * R.random(7), 0.5 + R.random(7)
LOC = np.array([0.0, 3.0])
PREC = np.linalg.inv(np.array([[1.0, 1.0], [1.0, 4.0]]))
A6 = R.normal(size=(4, 6))
Y4 = R.normal(size=4)


def student_t(q):
    return (-0.5 * (NU + 1.0) * np.log1p((q / SC) ** 2 / NU)).sum()


def mvn(y):
    return -0.5 * (y - LOC) @ (PREC @ (y - LOC))


def funnel(q):
    v, x = q[0], q[1:]
    return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()


def regression(q):  # a small linear model with a matrix captured from the closure, sigma = exp(q[-1])
    w, ls = q[:-1], q[-1]
    r = Y4 - A6[:, :5] @ w
    return -0.5 * np.sum(r * r) * np.exp(-2.0 * ls) - 4 * ls - 0.5 * np.dot(w, w) - 0.5 * ls**2


def kitchen_sink(q):
    a = np.tanh(q[0]) + np.sin(q[1]) * np.cos(q[2]) + np.sqrt(1.0 + np.square(q[3])) + np.expm1(-np.abs(q[4]))
    b = tracing.where(q[0] > 0.1, q[0] ** 3, -q[0]) + np.maximum(q[1], q[2]) + np.minimum(q[3], 0.3) + np.logaddexp(q[4], q[0])
    c = tracing.softplus(q[1]) + 2.0 ** q[2] + np.power(1.5 + q[3] ** 2, 0.3) + (1.0 / (2.0 + q[4] ** 2)) + np.mean(q) + sum(q[1:3])
    return a + b - c + (q @ q) * 0.1


def gamma_mixture(q):  # Gamma(shape = exp(q[0]), rate = exp(q[1])) observations: lgamma of a traced shape parameter
    from scipy.special import gammaln
    a, b = np.exp(q[0]), np.exp(q[1])
    x = np.exp(q[2:])
    return np.sum(a * q[1] - gammaln(a) + (a - 1.0) * q[2:] - b * x) + q[2:].sum() - 0.5 * (q[0] ** 2 + q[1] ** 2)


XL = R.normal(size=(300, 5)) / 2.0
YL = (R.random(300) < 0.5).astype(np.float64)


def logistic(q):  # a data matrix captured from the closure: z = X @ q is ONE node used twice (value and adjoint merged)
    z = XL @ q
    return (YL * z - tracing.softplus(z)).sum() - 0.5 * (q @ q) / 4.0


def bernoulli_expit(q):  # scipy.special.expit as the link function: log p = y log s + (1 - y) log(1 - s), s = expit(X q)
    from scipy.special import expit
    s = expit(XL @ q)
    return np.sum(YL * np.log(s) + (1.0 - YL) * np.log1p(-s)) - 0.5 * np.sum(q * q)


def more_functions(q):  # the later additions: arctan sinh cosh erfc log2 log10 exp2
    from scipy.special import erfc
    return (np.arctan(q[0] * q[1]) + 0.1 * np.sinh(q[2]) - 0.05 * np.cosh(q[3] - q[0]) + np.log(0.5 * erfc(q[4] * 0.7)) + np.log2(1.0 + q[1] ** 2)
            - np.log10(2.0 + q[2] ** 2) + 0.01 * np.exp2(q[3]) - 0.5 * np.sum(q * q))


def numpy_idioms(q):  # norm / clip / sign / var / std / diff / max / min / T / reshape / zeros_like / logaddexp.reduce / concatenate
    a = -np.linalg.norm(q) + np.sum(np.clip(q, -0.4, 0.6) ** 2) + np.sum(np.sign(q) * q) * 0.1 - np.var(q) - 0.3 * np.std(q[1:], ddof=1)
    b = -np.sum(np.diff(q) ** 2) + 0.2 * np.max(q) - 0.1 * q.min() + q.T @ q * 0.05 - np.sum((q.reshape(-1) - np.zeros_like(q)) ** 2) * 0.01
    c = np.logaddexp.reduce(q[:4]) - np.sum(np.concatenate([q[:2], q[4:]]) ** 2) * 0.07 + np.sum(np.stack([q[0], q[2]]) * np.array([0.3, -0.2]))
    return a + b + c


XS = R.normal(size=(50, 4))
YS = np.eye(3)[R.integers(0, 3, size=50)]


def softmax_regression(q):  # a matrix-valued parameter written as a list of its rows: W[k] = q[4 k : 4 k + 4], logits_k = X @ W[k]
    logits = [XS @ q[4 * k:4 * k + 4] for k in range(3)]
    picked = sum((YS[:, k] * logits[k]).sum() for k in range(3))
    return picked - np.sum(tracing.logsumexp(logits)) - 0.5 * np.sum(q * q) / 9.0


def shared_under_where(q):  # a shared sub-expression with a use inside a where-branch (never merged: see _RevGen.count_uses)
    u = np.exp(q[:3]) + q[3:6] ** 2
    return np.sum(tracing.where(q[:3] > 0.1, u * q[3:6], -u) + np.sin(u)) - 0.5 * np.sum(q * q)


GROUP = R.integers(0, 7, size=120)
YG = R.normal(size=120) + GROUP * 0.3


def hierarchical(q):  # random effects by group: theta[group] is a gather through a captured integer array
    mu, lt, theta = q[0], q[1], q[2:]
    tau = np.exp(lt)
    r = YG - theta[GROUP]
    return (-0.5 * mu * mu / 25.0 + lt - 0.5 * tau * tau / 4.0 - 7 * lt - 0.5 * np.sum((theta - mu) ** 2) / (tau * tau)
            - 0.5 * np.sum(r * r) + np.sum(np.tanh(q[2:][GROUP[:30]]) * YG[:30]))


def shared_in_comparison(q):  # a shared node whose other use is a comparison only (which passes no adjoint on)
    z = XL[:40] @ q
    return np.sum(tracing.where(z > 0.1, YL[:40], -YL[:40]) * 0.3 + z * z) + np.sum(tracing.where(q > 0.0, 1.0, 2.0) * q)


OBS = R.normal(size=60) * 2.0


def mixture(q):  # three-component Gaussian mixture: means q[0:3], log-scales q[3:6], unnormalised log-weights q[6:9]
    mu, ls, lw = q[0:3], q[3:6], q[6:9]
    comp = [lw[k] - ls[k] - 0.5 * ((OBS - mu[k]) * np.exp(-ls[k])) ** 2 for k in range(3)]   # per observation, component k
    return np.sum(tracing.logsumexp(comp)) - 60 * tracing.logsumexp(lw) - 0.5 * np.sum(q * q) / 9.0


CASES = {"student_t": (student_t, 7, True), "hierarchical": (hierarchical, 9, False), "mixture": (mixture, 9, False), "shared_in_comparison": (shared_in_comparison, 5, False), "gamma": (gamma_mixture, 6, False), "logistic": (logistic, 5, False), "bernoulli_expit": (bernoulli_expit, 5, False), "more_functions": (more_functions, 5, False), "numpy_idioms": (numpy_idioms, 6, False), "softmax_regression": (softmax_regression, 12, False),
         "shared_under_where": (shared_under_where, 6, False), "mvn": (mvn, 2, False), "funnel": (funnel, 10, False),
         "regression": (regression, 6, False), "kitchen_sink": (kitchen_sink, 5, False)}

HARNESS = r"""
#include <cstdio>
#include <cmath>
#define __device__
#include "dual.cuh"
using aehmc::Dual;
%(source)s
template <class T> struct Row {
  const double *q; int seed, D;
  T operator[](int i) const;
  int size() const { return D; }
};
template <> double Row<double>::operator[](int i) const { return q[i]; }
template <> Dual Row<Dual>::operator[](int i) const { return Dual(q[i], i == seed ? 1.0 : 0.0); }
%(params)s
int main() {
  const int D = %(D)d;
  const double q[] = {%(q)s};
#if %(elem)d
  double v = 0.0;
  for (int i = 0; i < D; i++) v += aehmc_logp(q[i], (long long)i, prm);
  std::printf("%%.17g\n", v);
  for (int i = 0; i < D; i++) std::printf("%%.17g\n", aehmc_logp(Dual(q[i], 1.0), (long long)i, prm).d);
#else
  Row<double> r{q, 0, D};
  std::printf("%%.17g\n", (double)aehmc_logp(r, prm));
  for (int i = 0; i < D; i++) { Row<Dual> rd{q, i, D}; std::printf("%%.17g\n", aehmc_logp(rd, prm).d); }
#endif
  return 0;
}
"""


REV_HARNESS = r"""
#include <cstdio>
#include <cmath>
#define __device__
#define AEHMC_LANES 1
#define AEHMC_WSUM(x) (x)
#define AEHMC_ATOMIC_ADD(p, v) (*(p) += (v))
#define AEHMC_SYNC()
#include "dual.cuh"
%(source)s
%(params)s
int main() {
  const int D = %(D)d;
  const double q[] = {%(q)s};
  double g[%(D)d] = {0};
  std::printf("%%.17g\n", aehmc_logp_grad(q, g, 0, prm));
  for (int i = 0; i < D; i++) std::printf("%%.17g\n", g[i]);
  return 0;
}
"""


def params_decl(tr):
    params = "".join(f"static const double prm{k}[] = {{{', '.join(repr(float(x)) for x in p)}}};\n" for k, p in enumerate(tr.params))
    return params + "static const double *const prm[] = {" + ", ".join([f"prm{k}" for k in range(len(tr.params))] + ["nullptr"]) + "};\n"


def run_cpp_reverse(tr, q, tmp_path, name):
    """the reverse-mode program (one lane: the loops run whole, the wavefront reductions are identities)"""
    src = tmp_path / f"{name}_rev.cpp"
    src.write_text(REV_HARNESS % dict(source=tr.grad_source, params=params_decl(tr), D=len(q), q=", ".join(repr(float(x)) for x in q)))
    exe = tmp_path / f"{name}_rev"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = [float(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    return out[0], np.array(out[1:])


def run_cpp(tr, q, tmp_path, name):
    src = tmp_path / f"{name}.cpp"
    src.write_text(HARNESS % dict(source=tr.source, params=params_decl(tr), D=len(q), q=", ".join(repr(float(x)) for x in q), elem=int(tr.elementwise)))
    exe = tmp_path / name
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "aehmc_amd", "csrc"), "-o", str(exe), str(src)])
    out = [float(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    return out[0], np.array(out[1:])


@pytest.mark.parametrize("name", sorted(CASES))
def test_traced_density_and_gradient_match_the_python_function(name, tmp_path):
    fn, D, elementwise = CASES[name]
    tr = tracing.trace(fn, D)
    assert tr.elementwise == elementwise and tr.dim == D
    for trial in range(2):
        q = 0.7 * np.random.default_rng(trial).normal(size=D)
        v, g = run_cpp(tr, q, tmp_path, f"{name}{trial}")
        assert v == pytest.approx(float(fn(q)), rel=1e-13, abs=1e-13)
        h = 1e-6
        fd = np.array([(fn(q + h * np.eye(D)[i]) - fn(q - h * np.eye(D)[i])) / (2 * h) for i in range(D)])
        np.testing.assert_allclose(g, fd, rtol=2e-7, atol=2e-8)
        if not tr.elementwise:  # the reverse sweep (taken above 64 coordinates) == the forward passes
            vr, gr = run_cpp_reverse(tr, q, tmp_path, f"{name}{trial}")
            assert vr == pytest.approx(v, rel=1e-14, abs=1e-14)
            np.testing.assert_allclose(gr, g, rtol=1e-12, atol=1e-13)


def random_density(seed, D):
    """a random joint density: hyper-parameters at the head of the position, vector terms over slices of the rest, nested
    reductions, where / maximum, captured arrays"""
    r = np.random.default_rng(seed)
    n = D - 3
    w, c = r.normal(size=n), 0.5 + r.random(n)
    A = r.normal(size=(3, n)) / np.sqrt(n)
    picks = r.integers(0, 9, size=4)
    G = max(2, n // 3)
    grp = r.integers(0, G, size=2 * n)
    w2 = r.normal(size=2 * n)
    B = r.normal(size=(n, n)) / n

    def fn(q):
        a, b, s = q[0], q[1], q[2]
        x = q[3:]
        terms = [lambda: (-0.5 * (x - a) ** 2 * np.exp(-2.0 * s)).sum() - n * s,
                 lambda: -np.sum(np.log1p(np.square((x - w) / c))) * (1.0 + b * b) ** 0.5,
                 lambda: np.tanh(a) * (x[: n // 2] * x[n - n // 2:]).sum() - 0.5 * np.dot(x, x) / (1.0 + tracing.softplus(b)),
                 lambda: -0.5 * np.sum((A @ x - np.array([a, b, s])) ** 2),
                 lambda: np.sum(tracing.where(x > a, np.sin(x) * b, -np.abs(x - a))) * 0.3 - np.logaddexp(a, b),
                 lambda: -(np.maximum(x, w) * c).sum() * np.exp(-np.abs(s)) + (x[1:] - x[:-1]).mean() * np.cos(a),
                 lambda: -0.5 * np.sum((w2 - x[:G][grp] * np.exp(0.1 * b)) ** 2) / (2 * n),            # a gather (random effects by group)
                 lambda: -0.5 * np.dot(x, B @ x) * (1.0 + 0.1 * np.tanh(s)) if n > 32 else -0.5 * np.dot(x - a, B @ (x - a)),
                 lambda: np.exp(-np.sum(x * x) / n) * a - np.log1p(np.sum(np.square(x - b)) / n)]     # functions of reductions
        out = -0.5 * (a * a + b * b + s * s)
        for k in picks:
            out = out + terms[k]()
        return out

    return fn


def test_a_long_reduction_that_cannot_be_spread_over_the_lanes_warns():
    X = np.random.default_rng(0).normal(size=(2000, 80))   # 80 coefficients: the inner reduction exceeds the private accumulators
    with pytest.warns(UserWarning, match="runs on ONE lane"):
        tracing.trace(lambda q: -0.5 * np.sum((X @ q) ** 2) - 0.5 * (q @ q), 80)


def test_gather_index_out_of_range_is_an_indexerror():
    with pytest.raises(IndexError, match="out of range"):
        tracing.trace(lambda q: q[np.array([0, 5])].sum(), 3)


@pytest.mark.parametrize("seed", range(18))
def test_reverse_mode_equals_forward_mode_on_random_densities(seed, tmp_path):
    """forward- and reverse-mode gradients equal to 1e-12 on random densities (both compiled as plain
    C++: the Dual template against the reverse sweep), and both equal to central differences of the Python function"""
    D = [9, 17, 70][seed % 3]
    fn = random_density(seed, D)
    tr = tracing.trace(fn, D)
    assert not tr.elementwise and "aehmc_logp_grad" in tr.grad_source
    q = 0.6 * np.random.default_rng(100 + seed).normal(size=D)
    v, g = run_cpp(tr, q, tmp_path, f"rnd{seed}")
    vr, gr = run_cpp_reverse(tr, q, tmp_path, f"rnd{seed}")
    assert v == pytest.approx(float(fn(q)), rel=1e-12, abs=1e-12) and vr == pytest.approx(v, rel=1e-13, abs=1e-13)
    np.testing.assert_allclose(gr, g, rtol=1e-12, atol=1e-12)
    h = 1e-6
    fd = np.array([(fn(q + h * np.eye(D)[i]) - fn(q - h * np.eye(D)[i])) / (2 * h) for i in range(D)])
    np.testing.assert_allclose(gr, fd, rtol=5e-7, atol=5e-7)


def test_readme_function_is_the_builtin_standard_normal_expression():
    """README.md:27-36: logprob of N(0, 1) at a scalar position.  The emitted expression is, operation for operation,
    the built-in StdNormal target's (engine.cuh: u = 0.5 (q q) + log sqrt(2 pi), g = q), so the README value is
    reproduced bit for bit on the GPU (tests/test_gpu_callable.py)."""
    tr = tracing.trace(lambda y: -0.5 * y**2 - 0.5 * np.log(2 * np.pi), 1, scalar=True)
    assert tr.elementwise and tr.params == []
    assert "return T(((-0.5 * square(q)) - 0.9189385332046727));" in tr.source
    # the same function handed a vector of one entry (a [C, 1] position) takes the same form
    assert tracing.trace(lambda y: -0.5 * y**2 - 0.5 * np.log(2 * np.pi), 1).source == tr.source


def test_targets_from_callable_picks_the_target_class():
    t1 = targets.from_callable(student_t, 7)
    assert isinstance(t1, targets.Custom) and not t1.hand_gradient and len(t1.param_list) == 3 and t1.dim == 7
    t2 = targets.from_callable(mvn, 2)
    assert isinstance(t2, targets.CustomJoint) and t2.dim == 2 and "aehmc_logp" in t2.source
    assert targets.as_target(t2, 2) is t2
    assert targets.as_target(mvn, 2) is targets.as_target(mvn, 2)  # traced once per function and shape
    with pytest.raises(TypeError, match="Target or a Python function"):
        targets.as_target(3.0, 2)


@pytest.mark.parametrize("fn, match", [
    (lambda q: q.sum() if q[0] > 0 else 0.0, "control flow cannot be traced"),
    (lambda q: __import__("math").exp(q[0]), "use the numpy functions"),
    (lambda q: np.arcsinh(q).sum(), "numpy.arcsinh is not supported"),
    (lambda q: q * 2.0, "must return a scalar"),
    (lambda q: 1.0, "does not depend on the position"),
    (lambda q: q[np.array([0.5, 1.0])].sum(), "integer arrays"),
    (lambda q: (q > 0).sum(), "comparison"),
    (lambda q: np.cumsum(q)[-1], "numpy.cumsum is not supported"),
    (lambda q: (q[:2] + q).sum(), "do not broadcast"),
    (lambda q: (np.ones((2, 2, 2)) * q[0]).sum(), "dimensions are not supported"),
    (lambda q: np.max(np.concatenate([q] * 22)), "supported up to 64"),
    (lambda q: q.reshape(3, 1).sum(), "traced values are scalars and vectors"),
    (lambda q: np.stack([q, q]).sum(), "would be a matrix"),
    (lambda q: q.astype(int).sum(), "astype"),
    (lambda q: np.linalg.norm(q, ord=1), "numpy.norm is not supported"),
])
def test_untraceable_operations_raise_typeerror_at_trace_time(fn, match):
    with pytest.raises(TypeError, match=match):
        tracing.trace(fn, 3)


def test_max_over_a_long_vector_is_refused_with_its_length():
    with pytest.raises(TypeError, match="max / min over 100 traced entries"):
        tracing.trace(lambda q: np.max(q), 100)
