"""``logprob_fn`` as a Python function of the position -- what the reference takes (README.md:27-36, aehmc/hmc.py:16-40) --
traced once (aehmc_amd/tracing.py), compiled with hipRTC and differentiated by the engine (csrc/dual.cuh).
Parity: the README value bit for bit; whole transitions against the numpy restatement (oracle/np_oracle.py) driven by
THE SAME Python function on plain numpy arrays with central-difference-free analytic gradients; statistics as
/root/reference/tests/test_hmc.py:190-264 with the model written as a Python function."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import np_oracle as no  # noqa: E402

RTOL = 1e-9


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


def test_readme_example_with_a_python_logprob_fn_is_bit_exact():
    """README.md:22-54 (config c1): RandomStream(seed=0), logprob of N(0, 1) written as a Python function, y0 = 0,
    step size 1e-2, unit inverse mass matrix -> 1.1034719409361107."""
    from aehmc_amd import RandomStream, nuts
    logprob_fn = lambda y: -0.5 * y**2 - 0.5 * np.log(2 * np.pi)  # noqa: E731
    kernel = nuts.new_kernel(RandomStream(seed=0), logprob_fn)
    state = nuts.new_state(0.0, logprob_fn)
    assert state.potential_energy.item() == 0.5 * np.log(2 * np.pi) and state.potential_energy_grad.item() == 0.0
    info, _ = kernel(state, 1e-2, 1.0)
    assert info.state.position.item() == 1.1034719409361107
    assert info.num_doublings.item() == 8 and info.n_leapfrog.item() == 136 and not info.is_diverging.item()
    assert info.acceptance_probability.item() == pytest.approx(0.9999767760191554, rel=1e-12)


LOC, SCALE, RHO = np.array([0.0, 3.0]), np.array([1.0, 2.0]), 0.5
COV = np.diag(SCALE**2)
COV[0, 1] = COV[1, 0] = RHO * SCALE[0] * SCALE[1]
PREC = np.linalg.inv(COV)
PREC = 0.5 * (PREC + PREC.T)


def mvn_logprob(y):  # /root/reference/tests/test_hmc.py:170-187 (multivariate_normal_model), up to the constant
    d = y - LOC
    return -0.5 * d @ (PREC @ d)


def test_hmc_mcse_with_the_model_as_a_python_function():
    # /root/reference/tests/test_hmc.py:190-264: mu = [0, 3], sigma = [1, 2], rho = .5, eps = 1, L = 30, imm = sigma (sic)
    from scipy import stats
    from aehmc_amd import RandomStream, hmc
    C = 512
    kernel = hmc.new_kernel(RandomStream(seeds=[10_000 + c for c in range(C)]), mvn_logprob)
    q0 = np.random.default_rng(0).standard_normal((C, 2))
    state = hmc.new_state(dev(q0), mvn_logprob)
    _, info, _, _ = kernel.sample(state, 1.0, SCALE, 30, 300, keep_samples=False)   # burn-in
    samples, info, acc, div = kernel.sample(info.state._replace(momentum=None), 1.0, SCALE, 30, 400)
    s = samples.cpu().numpy()  # [400, C, 2]
    assert not div.any().item()

    def pvalue(delta):  # chains are independent: MCSE from the spread of per-chain means
        m = delta.mean(axis=0)
        return stats.norm.sf(np.abs(m.mean(axis=0)) / (m.std(axis=0, ddof=1) / np.sqrt(C)))

    assert np.all(pvalue(s - LOC) > 0.001)
    assert np.all(pvalue(np.square(s - LOC) - SCALE**2) > 0.001)
    assert np.all(pvalue(np.prod(s - LOC, axis=2) / np.prod(SCALE) - RHO) > 0.001)


class NumpyTarget:
    """The numpy restatement's target from the SAME Python function (called on plain arrays) and an analytic gradient."""

    def __init__(self, fn, grad):
        self.fn, self.grad = fn, grad

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        return float(-self.fn(q)), -np.asarray(self.grad(q), dtype=np.float64)


R = np.random.default_rng(8)
NU, SC = 3.0 + 5 * R.random(70), 0.5 + R.random(70)


def student_t(q):
    return (-0.5 * (NU + 1.0) * np.log1p((q / SC) ** 2 / NU)).sum()


def student_t_grad(q):
    z = q / SC
    return -(NU + 1.0) * z / (NU + z * z) / SC


def funnel(q):
    v, x = q[0], q[1:]
    return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()


def funnel_grad(q):
    v, x = q[0], q[1:]
    g = np.empty_like(q)
    g[0] = -v / 9.0 + 0.5 * np.sum(x * x) * np.exp(-v) - 0.5 * (len(q) - 1)
    g[1:] = -x * np.exp(-v)
    return g


def mvn_grad(y):
    return -(PREC @ (y - LOC))


MODELS = {"student_t": (student_t, student_t_grad, 70, "Custom"), "funnel": (funnel, funnel_grad, 10, "CustomJoint"),
          "funnel100": (funnel, funnel_grad, 100, "CustomJoint"), "mvn": (mvn_logprob, mvn_grad, 2, "CustomJoint")}


@pytest.mark.parametrize("model", sorted(MODELS))
@pytest.mark.parametrize("sampler", ["nuts", "hmc"])
def test_python_logprob_fn_transitions_match_numpy_restatement(model, sampler):
    from aehmc_amd import RandomStream, hmc, nuts, targets
    fn, grad, D, cls = MODELS[model]
    assert type(targets.as_target(fn, D)).__name__ == cls
    otgt = NumpyTarget(fn, grad)
    r = np.random.default_rng(len(model))
    C, n, eps = 4, 3, 0.1
    q0 = 0.5 * r.normal(size=(C, D))
    imm = 0.5 + r.random(D)
    seeds = [300 + c for c in range(C)]
    mod, omk, extra = (nuts, no.nuts_kernel, ()) if sampler == "nuts" else (hmc, no.hmc_kernel, (7,))
    kw = dict(max_num_expansions=5) if sampler == "nuts" else {}
    kern = mod.new_kernel(RandomStream(seeds=seeds), fn, **kw)
    state = mod.new_state(dev(q0), fn)
    okern = [omk(no.RandomStream(sd), otgt, **kw) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    for _ in range(n):
        info, _ = kern(state, eps, imm, *extra)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, imm, *extra)
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-10)
            assert bool(info.is_diverging[c]) == bool(o.is_diverging)
            if sampler == "nuts":
                assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
                assert bool(info.is_turning[c]) == bool(o.is_turning)


SCHOOLS_Y = np.array([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
SCHOOLS_SIGMA = np.array([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])


def schools(q, J):
    """eight schools, non-centred, tiled to J schools: q = [mu, log tau, eta_1..J] (tests/test_gpu_autodiff.py's model)"""
    y, sg = np.resize(SCHOOLS_Y, J), np.resize(SCHOOLS_SIGMA, J)
    mu, lt, eta = q[0], q[1], q[2:]
    tau = np.exp(lt)
    z = (y - (mu + tau * eta)) / sg
    return -0.5 * mu * mu / 25.0 - np.log1p(tau * tau / 25.0) + lt + (-0.5 * eta * eta - 0.5 * z * z).sum()


def schools_grad(q, J):
    y, sg = np.resize(SCHOOLS_Y, J), np.resize(SCHOOLS_SIGMA, J)
    mu, lt, eta = q[0], q[1], q[2:]
    tau = np.exp(lt)
    z = (y - (mu + tau * eta)) / sg
    g = np.empty_like(q)
    g[0] = -mu / 25.0 + np.sum(z / sg)
    g[1] = -(2.0 * tau * tau / 25.0) / (1.0 + tau * tau / 25.0) + 1.0 + np.sum(z * tau * eta / sg)
    g[2:] = -eta + z * tau / sg
    return g


@pytest.mark.parametrize("D", [65, 200, 1000, 2048])
@pytest.mark.parametrize("model", ["funnel", "schools"])
def test_reverse_mode_joint_densities_match_numpy_restatement(model, D):
    """above 64 coordinates a traced density is differentiated in ONE reverse sweep (the reference:
    aesara.grad, /root/reference/aehmc/hmc.py:33-34, integrators.py:61-65).  NUTS transitions against the numpy
    restatement driven by the same Python function and the analytic gradient: 1e-9, every discrete output identical."""
    from aehmc_amd import RandomStream, nuts, targets
    if model == "funnel":
        fn, grad = funnel, funnel_grad
    else:
        fn, grad = (lambda q: schools(q, D - 2)), (lambda q: schools_grad(q, D - 2))
    tgt = targets.as_target(fn, D)
    assert isinstance(tgt, targets.CustomJoint) and "#define AEHMC_JOINT_GRAD 1" in tgt.source
    otgt = NumpyTarget(fn, grad)
    r = np.random.default_rng(D)
    C, n, eps = 3, 2, (0.02 if model == "funnel" else 0.05)
    q0 = 0.3 * r.normal(size=(C, D))
    imm = 0.5 + r.random(D)
    seeds = [900 + c for c in range(C)]
    kern = nuts.new_kernel(RandomStream(seeds=seeds), fn, max_num_expansions=4)
    state = nuts.new_state(dev(q0), fn)
    for c in range(C):  # new_state: U and the whole gradient from one sweep
        U, g = otgt(q0[c])
        np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
        np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-11, atol=1e-12)
    okern = [no.nuts_kernel(no.RandomStream(sd), otgt, max_num_expansions=4) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    for _ in range(n):
        info, _ = kern(state, eps, imm)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, imm)
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-10)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)


def test_notebook_regression_written_as_a_python_function(regression_data):
    """examples/LinearRegression.ipynb:126-166 (w ~ N(0, 1), n ~ Gamma(2, 1), y_i ~ N(X_i w, n), sampled in q = [w, log n])
    written as a Python function over the 10^4 data rows: two coordinates, so the one-launch kernels take it -- with the
    reverse-mode program (AEHMC_JOINT_GRAD_SMALL: the row sum spread over the wavefront's lanes).  G3: logp([3, log 10]) =
    -32238.026021294307 (:188); G2: the notebook's single HMC step (:293-297); and the built-in LinearRegression target on
    the same data as a second opinion on the gradient."""
    from aehmc_amd import RandomStream, hmc, targets
    X, y = regression_data
    half_log_2pi = 0.5 * np.log(2 * np.pi)

    def logprob_fn(q):
        w, ls = q[0], q[1]
        n = np.exp(ls)
        r = y - X * w
        return (-0.5 * w * w - half_log_2pi) + (ls - n) + ls + (-0.5 * (r / n) ** 2 - ls - half_log_2pi).sum()

    tgt = targets.as_target(logprob_fn, 2)
    assert isinstance(tgt, targets.CustomJoint) and "#define AEHMC_JOINT_GRAD_SMALL 1" in tgt.source
    state = hmc.new_state(dev(np.array([3.0, np.log(10.0)])), logprob_fn)
    assert -state.potential_energy.item() == pytest.approx(-32238.026021294307, rel=1e-12)  # G3
    builtin = hmc.new_state(dev(np.array([3.0, np.log(10.0)])), targets.LinearRegression(X, y))
    np.testing.assert_allclose(state.potential_energy_grad.cpu().numpy(), builtin.potential_energy_grad.cpu().numpy(), rtol=1e-11)
    # G2: RandomStream(seed=0), start [3, log 0.21], eps = 5e-5, L = 1024, imm = [1, 1]
    kernel = hmc.new_kernel(RandomStream(seed=0), logprob_fn)
    info, _ = kernel(hmc.new_state(dev(np.array([3.0, np.log(0.21)])), logprob_fn), 5e-5, np.ones(2), 1024)
    np.testing.assert_allclose(info.state.position.cpu().numpy(), [2.99946192, -1.30494977], atol=5e-9)
    assert info.state.potential_energy.item() == pytest.approx(12433.00653542, abs=5e-8)
    np.testing.assert_allclose(info.state.potential_energy_grad.cpu().numpy(), [-489.93218536, -22571.36970197], atol=5e-8)
    assert info.acceptance_probability.item() == 1.0 and not info.is_diverging.item()


@pytest.mark.parametrize("N, D", [(63, 3), (65, 8), (2500, 8), (10000, 20)])
def test_logistic_regression_written_as_a_python_function(N, D):
    """A GLM written the way a user of the reference writes it -- z = X @ q over a captured data matrix, a sum over the
    rows, a Gaussian prior -- against targets.CustomGLM (the same model as HIP source, density only) and the numpy
    restatement with the analytic gradient.  The rows are spread over the lanes of the chain's wavefront, the D
    coefficient adjoints collected in per-lane accumulators (tracing._private_leaves)."""
    from aehmc_amd import RandomStream, nuts, targets, tracing
    r = np.random.default_rng(N + D)
    X = r.normal(size=(N, D)) / np.sqrt(D)
    y = (r.random(N) < 0.5).astype(np.float64)

    def logprob_fn(q):
        z = X @ q
        return (y * z - tracing.softplus(z)).sum() - 0.5 * (q @ q) / 4.0

    def grad(q):
        return X.T @ (y - 1.0 / (1.0 + np.exp(-(X @ q)))) - q / 4.0

    tgt = targets.as_target(logprob_fn, D)
    assert isinstance(tgt, targets.CustomJoint) and ("AEHMC_JOINT_GRAD_SMALL" in tgt.source) == (N >= 256)
    glm = targets.CustomGLM("""
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
""", dev(X), dev(y))
    C, eps = 4, 0.3 / np.sqrt(N)
    q0 = 0.3 * r.normal(size=(C, D))
    s1, s2 = nuts.new_state(dev(q0), logprob_fn), nuts.new_state(dev(q0), glm)
    np.testing.assert_allclose(s1.potential_energy.cpu().numpy(), s2.potential_energy.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(s1.potential_energy_grad.cpu().numpy(), s2.potential_energy_grad.cpu().numpy(), rtol=1e-10, atol=1e-11)
    otgt = NumpyTarget(logprob_fn, grad)
    seeds = [40 + c for c in range(C)]
    kern = nuts.new_kernel(RandomStream(seeds=seeds), logprob_fn, max_num_expansions=5)
    okern = [no.nuts_kernel(no.RandomStream(sd), otgt, max_num_expansions=5) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    state = s1
    for _ in range(3):
        info, _ = kern(state, eps, np.ones(D))
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, np.ones(D))
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)


@pytest.mark.parametrize("G, N", [(10, 1000), (200, 5000)])
def test_hierarchical_model_with_a_gather_as_a_python_function(G, N):
    """Random effects by group -- theta[group] with `group` a captured integer array (a gather; its adjoint is a scatter
    with LDS atomics inside the loop over the observations) -- with G + 2 coordinates: 12 (forward-mode kernels' size, but
    the reverse-mode program is taken for the 1000-term reductions) and 202 (reverse mode on the joint-rows kernels)."""
    from aehmc_amd import RandomStream, nuts, targets
    r = np.random.default_rng(G)
    group = r.integers(0, G, size=N)
    y = r.normal(size=N) + 0.5 * r.normal(size=G)[group]

    def logprob_fn(q):
        mu, lt, theta = q[0], q[1], q[2:]
        tau2 = np.exp(2.0 * lt)
        res = y - theta[group]
        return -0.5 * mu * mu / 25.0 + lt - 0.5 * tau2 / 4.0 - G * lt - 0.5 * np.sum((theta - mu) ** 2) / tau2 - 0.5 * np.sum(res * res)

    def grad(q):
        mu, lt, theta = q[0], q[1], q[2:]
        tau2 = np.exp(2.0 * lt)
        res = y - theta[group]
        g = np.empty_like(q)
        g[0] = -mu / 25.0 + np.sum(theta - mu) / tau2
        g[1] = 1.0 - tau2 / 4.0 - G + np.sum((theta - mu) ** 2) / tau2
        g[2:] = -(theta - mu) / tau2 + np.bincount(group, weights=res, minlength=G)
        return g

    D = G + 2
    tgt = targets.as_target(logprob_fn, D)
    assert isinstance(tgt, targets.CustomJoint) and "AEHMC_ATOMIC_ADD(&g[" in tgt.source
    otgt = NumpyTarget(logprob_fn, grad)
    C, eps = 3, 0.02
    q0 = 0.3 * r.normal(size=(C, D))
    state = nuts.new_state(dev(q0), logprob_fn)
    for c in range(C):
        U, g = otgt(q0[c])
        np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
        np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-10, atol=1e-10)
    seeds = [5 + c for c in range(C)]
    kern = nuts.new_kernel(RandomStream(seeds=seeds), logprob_fn, max_num_expansions=4)
    okern = [no.nuts_kernel(no.RandomStream(sd), otgt, max_num_expansions=4) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    for _ in range(2):
        info, _ = kern(state, eps, np.ones(D))
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, np.ones(D))
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)


def test_workgroup_per_chain_kernels_agree_with_the_wavefront_per_chain_ones(regression_data):
    """A traced density with long data sweeps and few chains runs with a WORKGROUP per chain (k_nuts_joint_wg /
    k_hmc_joint_wg: eight wavefronts run the generated program together); engine option joint_wg = 0 keeps the wavefront
    per chain.  Same transitions: discrete outputs identical, values to rounding (the sums are associated differently)."""
    from aehmc_amd import RandomStream, hmc, nuts
    from aehmc_amd.engine import get_engine
    X, y = regression_data
    h = 0.5 * np.log(2 * np.pi)

    def logprob_fn(q):
        w, ls = q[0], q[1]
        n = np.exp(ls)
        r = y - X * w
        return (-0.5 * w * w - h) + (ls - n) + ls + (-0.5 * (r / n) ** 2 - ls - h).sum()

    eng = get_engine()
    C = 6
    q0 = np.array([3.0, 0.0]) + 0.01 * np.random.default_rng(3).normal(size=(C, 2))
    imm = np.array([1e-4, 0.5e-4])
    out = {}
    try:
        for mode in (0, 2):
            eng.set_option("joint_wg", mode)
            kern = nuts.new_kernel(RandomStream(seeds=range(C)), logprob_fn, max_num_expansions=6)
            state = nuts.new_state(dev(q0), logprob_fn)
            samples, info, acc, div = kern.sample(state, 0.5, imm, 4)
            hk = hmc.new_kernel(RandomStream(seeds=range(C)), logprob_fn)
            hs, hinfo, hacc, hdiv = hk.sample(hmc.new_state(dev(q0), logprob_fn), 0.3, imm, 9, 3)
            out[mode] = (samples.cpu().numpy(), info.n_leapfrog.cpu().numpy(), acc.cpu().numpy(), hs.cpu().numpy(), hacc.cpu().numpy())
    finally:
        eng.set_option("joint_wg", 1)
    assert np.array_equal(out[0][1], out[2][1]) and out[0][1].sum() > 4 * C
    for k in (0, 2, 3, 4):
        np.testing.assert_allclose(out[0][k], out[2][k], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AEHMC_CROSSCHECK_SEEDS", "8"))))
def test_generated_reverse_mode_programs_agree_with_forward_mode_on_the_device(seed):
    """Random joint densities (tests/test_tracing.py: hyper-parameters, slices, nested reductions, where / maximum, a
    captured matrix) run three ways on the GPU: forward mode (dual numbers in the lanes / row passes), the generated
    reverse-mode program on a wavefront per chain, and the same program on a workgroup per chain.  The gradients are
    different programs over the same expression: trajectories agree at 1e-9, discrete outputs are identical.
    (Eight seeds in the suite -- each compiles four to five programs, ~10 s on a cold hipRTC cache; AEHMC_CROSSCHECK_SEEDS=15
    is the run recorded in profiles/r6/gpu_suite_cold_durations.txt.)"""
    from test_tracing import random_density
    D = [9, 17, 70, 40, 150][seed % 5]
    three_way(random_density(seed, D), D, seed)


@pytest.mark.parametrize("name", ["mixture", "hierarchical", "gamma", "kitchen_sink", "bernoulli_expit", "more_functions", "numpy_idioms", "softmax_regression"])
def test_named_models_three_ways_on_the_device(name):
    """tests/test_tracing.py's models (a three-component Gaussian mixture through logsumexp, random effects with a gather,
    Gamma observations with a traced shape parameter -- lgamma / digamma --, every supported function at once, the later
    additions expit / arctan / sinh / cosh / erfc / log2 / log10 / exp2): forward
    mode, reverse mode and a workgroup per chain on the GPU"""
    import test_tracing
    fn, D, _ = test_tracing.CASES[name]
    three_way(fn, D, len(name))


def three_way(fn, D, seed):
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    C = 5
    q0 = 0.4 * np.random.default_rng(50 + seed).normal(size=(C, D))
    imm = 0.5 + np.random.default_rng(seed).random(D)
    out = {}
    try:
        for name, rev, wg in (("forward", False, 0), ("reverse", True, 0), ("workgroup", True, 2)):
            eng.set_option("joint_wg", wg)
            tgt = targets.from_callable(fn, D, reverse=rev)
            kern = nuts.new_kernel(RandomStream(seeds=[7 + c for c in range(C)]), tgt, max_num_expansions=5)
            state = nuts.new_state(dev(q0), tgt)
            samples, info, acc, div = kern.sample(state, 0.05, imm, 3)
            out[name] = (state.potential_energy_grad.cpu().numpy(), samples.cpu().numpy(), acc.cpu().numpy(),
                         info.n_leapfrog.cpu().numpy(), div.cpu().numpy())
    finally:
        eng.set_option("joint_wg", 1)
    for name in ("reverse", "workgroup"):
        np.testing.assert_allclose(out[name][0], out["forward"][0], rtol=1e-11, atol=1e-11)
        assert np.array_equal(out[name][3], out["forward"][3]) and np.array_equal(out[name][4], out["forward"][4])
        np.testing.assert_allclose(out[name][1], out["forward"][1], rtol=RTOL, atol=1e-10)
        np.testing.assert_allclose(out[name][2], out["forward"][2], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("D, per_chain", [(65, False), (128, True), (200, False), (512, False), (512, True)])
def test_register_resident_joint_kernel_equals_the_rows_kernel_bitwise(D, per_chain):
    """64 < D <= 512: a traced joint density runs NUTS on the register-resident kernel (the chain in registers, the
    position handed to the generated program through LDS rows); engine option joint_resident = 0 keeps the one-launch
    kernel over the chains' L2 rows.  The leapfrog's arithmetic and the program are the same: the same bits, for single
    transitions and for sample()."""
    from aehmc_amd import PerChain, RandomStream, nuts
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    C = 9
    q0 = 0.3 * np.random.default_rng(D).normal(size=(C, D))
    imm = 0.5 + np.random.default_rng(D + 1).random(D)
    eps = 0.04
    if per_chain:
        imm = PerChain(dev(0.5 + np.random.default_rng(D + 1).random((C, D))))
        eps = PerChain(dev(0.03 + 0.02 * np.random.default_rng(D + 2).random(C)))
    out = {}
    try:
        for mode in (1, 0):
            eng.set_option("joint_resident", mode)
            kern = nuts.new_kernel(RandomStream(seeds=[11 + c for c in range(C)]), funnel, max_num_expansions=6)
            state = nuts.new_state(dev(q0), funnel)
            info, _ = kern(state, eps, imm)
            samples, info2, acc, div = kern.sample(info.state._replace(momentum=None), eps, imm, 3)
            out[mode] = (info.state.position, info.n_leapfrog, info.acceptance_probability, samples, acc, info2.n_leapfrog,
                         kern._nuts["holder"]["rng"].clone())
    finally:
        eng.set_option("joint_resident", 1)
    assert int(out[1][1].sum()) > C
    for a, b in zip(out[1], out[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D, per_chain", [(65, False), (128, True), (200, False), (512, True), (1000, False), (1000, True)])
def test_hmc_fused_kernel_with_a_traced_joint_density_equals_the_rows_kernel_bitwise(D, per_chain):
    """HMC, 64 < D <= 1024: k_hmc_fused compiled against the traced program (chain in registers, position / gradient
    rows of the program in LDS) against k_hmc_joint_rows (joint_resident = 0): the same bits for a single transition,
    for sample(), and for the generator state afterwards."""
    from aehmc_amd import PerChain, RandomStream, hmc
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    C = 9
    q0 = 0.3 * np.random.default_rng(D).normal(size=(C, D))
    imm = 0.5 + np.random.default_rng(D + 1).random(D)
    eps = 0.03
    if per_chain:  # what per-chain window adaptation hands back: a step size and a diagonal metric per chain
        imm = PerChain(dev(0.5 + np.random.default_rng(D + 1).random((C, D))))
        eps = PerChain(dev(0.02 + 0.02 * np.random.default_rng(D + 2).random(C)))
    out = {}
    try:
        for mode in (1, 0):
            eng.set_option("joint_resident", mode)
            kern = hmc.new_kernel(RandomStream(seeds=[11 + c for c in range(C)]), funnel)
            state = hmc.new_state(dev(q0), funnel)
            info, _ = kern(state, eps, imm, 7)
            samples, info2, acc, div = kern.sample(info.state._replace(momentum=None), eps, imm, 5, 4)
            info0, _ = kern(info2.state._replace(momentum=None), eps, imm, 0)  # (no leapfrog at all: the state comes back, accepted)
            out[mode] = (info.state.position, info.state.potential_energy, info.state.potential_energy_grad,
                         info.acceptance_probability, info.state.momentum, samples, acc, div, info2.state.potential_energy_grad,
                         info0.state.position, info0.state.potential_energy, info0.acceptance_probability,
                         kern._hmc["holder"]["rng"].clone())
            assert torch.equal(info0.state.position, info2.state.position) and float(info0.acceptance_probability.min()) == 1.0
    finally:
        eng.set_option("joint_resident", 1)
    assert 0.0 < float(out[1][6].min()) < 1.0  # (the history holds transitions that were not certain to be accepted)
    for a, b in zip(out[1], out[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D, full", [(10, False), (100, False), (12, True)])
def test_python_logprob_fn_under_window_adaptation_and_sample(D, full):
    """window_adaptation.run with a traced density: forward mode (D = 10), the reverse-mode program on the joint-rows
    kernels (D = 100), and a dense metric per chain (is_mass_matrix_full); a Gaussian with known moments so that the adapted
    metric and the samples can be checked."""
    from aehmc_amd import RandomStream, nuts, window_adaptation
    C = 128
    sd = 0.5 + np.arange(D) % 3

    def logprob_fn(q):
        z = (q - 1.0) / sd
        return -0.5 * (z @ z) - 0.05 * np.sum(z[1:] * z[:-1])   # (weakly coupled neighbours: a joint density)

    kernel = nuts.new_kernel(RandomStream(seeds=range(C)), logprob_fn, max_num_expansions=7)
    state = nuts.new_state(dev(1.0 + 0.3 * np.random.default_rng(1).normal(size=(C, D))), logprob_fn)
    state, (step_size, imm), _ = window_adaptation.run(kernel, state, num_steps=150, is_mass_matrix_full=full)
    samples, info, acc, div = kernel.sample(state, step_size, imm, 60)
    s = samples.cpu().numpy().reshape(-1, D)
    assert samples.shape == (60, C, D) and np.isfinite(s).all() and not bool(div.any())
    assert 0.6 < float(acc.mean()) < 0.98
    np.testing.assert_allclose(s.mean(axis=0), 1.0, atol=0.25)
    np.testing.assert_allclose(s.std(axis=0) / sd, 1.0, atol=0.2)


def test_untraceable_python_logprob_fn_raises_typeerror_before_anything_is_compiled():
    from aehmc_amd import nuts
    with pytest.raises(TypeError, match="control flow cannot be traced"):
        nuts.new_state(dev(np.zeros((3, 4))), lambda q: q.sum() if q[0] > 0 else -q.sum())
