"""The user writes the log-DENSITY, the engine differentiates it.

The reference takes any ``logprob_fn`` and obtains the potential's gradient with ``aesara.grad``
(/root/reference/aehmc/hmc.py:33-34, integrators.py:61-65); its tests sample a joint density built by aeppl
(/root/reference/tests/test_hmc.py:170-264).  Here a density is a HIP function template over its arithmetic type,
instantiated with ``aehmc::Dual`` (csrc/dual.cuh, forward mode) inside the run-time compiled kernels:
``targets.CustomJoint`` (non-separable, D <= 64: Neal's funnel, eight schools), and the density-only forms of
``targets.Custom`` / ``targets.CustomGLM``.  Parity: the same expression with an ANALYTIC gradient as a numpy callable
through the numpy restatement (oracle/np_oracle.py), chain by chain on identical seeds, 1e-9 with every discrete output
identical; statistics as /root/reference/tests/test_hmc.py:190-264; a wrong hand-written gradient is an error."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import np_oracle as no  # noqa: E402

RTOL = 1e-9

FUNNEL = """
// Neal's funnel: v = q[0] ~ N(0, 3^2), x_i = q[i] ~ N(0, exp(v)) -- density only
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  auto v = q[0];
  auto lp = -v * v / 18.0;
  for (int i = 1; i < q.size(); i++) lp += -0.5 * q[i] * q[i] * exp(-v) - 0.5 * v;
  return lp;
}
"""


class Funnel:
    def __init__(self, D):
        self.D = D

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        v, x = q[0], q[1:]
        lp = -v * v / 18.0
        for xi in x:  # the device's order of accumulation
            lp += -0.5 * xi * xi * np.exp(-v) - 0.5 * v
        g = np.empty_like(q)
        g[0] = -v / 9.0 + 0.5 * np.sum(x * x) * np.exp(-v) - 0.5 * (self.D - 1)
        g[1:] = -x * np.exp(-v)
        return float(-lp), -g


EIGHT_SCHOOLS = """
// eight schools, non-centred: q = [mu, log tau, eta_1..8]; mu ~ N(0, 5^2), tau ~ half-Cauchy(5) (+ log-Jacobian of
// tau = exp(q[1])), eta_j ~ N(0, 1), y_j ~ N(mu + tau eta_j, sigma_j^2); prm[0] = y, prm[1] = sigma
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  const double *y = prm[0], *sigma = prm[1];
  auto mu = q[0];
  auto tau = exp(q[1]);
  auto lp = -0.5 * mu * mu / 25.0 - log1p(tau * tau / 25.0) + q[1];
  for (int j = 0; j < 8; j++) {
    auto eta = q[2 + j];
    auto z = (y[j] - (mu + tau * eta)) / sigma[j];
    lp += -0.5 * eta * eta - 0.5 * z * z;
  }
  return lp;
}
"""
SCHOOLS_Y = np.array([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
SCHOOLS_SIGMA = np.array([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])


class EightSchools:
    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        mu, lt, eta = q[0], q[1], q[2:]
        tau = np.exp(lt)
        lp = -0.5 * mu * mu / 25.0 - np.log1p(tau * tau / 25.0) + lt
        z = (SCHOOLS_Y - (mu + tau * eta)) / SCHOOLS_SIGMA
        for j in range(8):
            lp += -0.5 * eta[j] * eta[j] - 0.5 * z[j] * z[j]
        g = np.empty_like(q)
        g[0] = -mu / 25.0 + np.sum(z / SCHOOLS_SIGMA)
        g[1] = -(2.0 * tau * tau / 25.0) / (1.0 + tau * tau / 25.0) + 1.0 + np.sum(z * tau * eta / SCHOOLS_SIGMA)
        g[2:] = -eta + z * tau / SCHOOLS_SIGMA
        return float(-lp), -g


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


@pytest.fixture()
def eng():
    from aehmc_amd.engine import get_engine
    e = get_engine()
    try:
        yield e
    finally:
        for name, val in (("resident_nuts", 2), ("fused_hmc", 1), ("fp_contract", 0)):
            e.set_option(name, val)


def models():
    from aehmc_amd import targets
    return {"funnel": (lambda: targets.CustomJoint(FUNNEL, dim=10), Funnel(10), 10),
            "schools": (lambda: targets.CustomJoint(EIGHT_SCHOOLS, dim=10, params=[SCHOOLS_Y, SCHOOLS_SIGMA]), EightSchools(), 10)}


def make_metric(kind, D, r):
    if kind == "scalar":
        return np.float64(0.7)
    if kind == "diag":
        return 0.5 + r.random(D)
    A = r.normal(size=(D, D))
    M = A @ A.T / D + np.eye(D)
    return 0.5 * (M + M.T)


@pytest.mark.parametrize("model", ["funnel", "schools"])
def test_joint_density_value_and_gradient(eng, model):
    """new_state: U = -logp and dU/dq from ONE dual evaluation per lane against the analytic gradient"""
    from aehmc_amd import nuts
    make, otgt, D = models()[model]
    r = np.random.default_rng(3)
    q0 = 0.7 * r.normal(size=(37, D))
    state = nuts.new_state(dev(q0), make())
    for c in range(q0.shape[0]):
        U, g = otgt(q0[c])
        np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
        np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("model,metric", [("funnel", "diag"), ("funnel", "dense"),
                                          ("schools", "diag"), ("schools", "dense")])
def test_joint_density_nuts_matches_numpy(eng, model, metric):
    """NUTS on the single-launch kernel compiled against the user's density: every transition against the numpy
    restatement with the analytic gradient, chain by chain on identical seeds"""
    from aehmc_amd import RandomStream, nuts
    make, otgt, D = models()[model]
    r = np.random.default_rng(len(model) + len(metric))
    C, n, max_exp, eps = 5, 4, 6, 0.12
    q0 = 0.5 * r.normal(size=(C, D))
    imm = make_metric(metric, D, r)
    seeds = [70 + c for c in range(C)]
    tgt = make()
    kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    okern = [no.nuts_kernel(no.RandomStream(sd), otgt, max_num_expansions=max_exp) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    for _ in range(n):
        info, _ = kern(state, eps, dev(imm) if metric == "dense" else imm)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, imm)
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)


@pytest.mark.parametrize("model,metric", [("funnel", "diag"), ("schools", "dense")])
def test_joint_density_hmc_and_sample_match_numpy(eng, model, metric):
    """HMC, and kernel.sample(n) (all transitions in one launch) == the same transitions one by one in the oracle"""
    from aehmc_amd import RandomStream, hmc
    make, otgt, D = models()[model]
    r = np.random.default_rng(11)
    C, n, L, eps = 4, 3, 9, 0.08
    q0 = 0.5 * r.normal(size=(C, D))
    imm = make_metric(metric, D, r)
    seeds = [200 + c for c in range(C)]
    tgt = make()
    kern = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    samples, info = kern.sample(hmc.new_state(dev(q0), tgt), eps, dev(imm) if metric == "dense" else imm, L, n)[:2]
    okern = [no.hmc_kernel(no.RandomStream(sd), otgt) for sd in seeds]
    for c in range(C):
        ost = no.new_state(q0[c].copy(), otgt)
        for t in range(n):
            o = okern[c](ost, eps, imm, L)
            ost = o.state._replace(momentum=None)
            np.testing.assert_allclose(samples[t, c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(info.state.potential_energy[c].item(), ost.potential_energy, rtol=RTOL)
        np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)


STUDENT_T_LOGP = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const double nu = prm[0][i], s = prm[1][i];   // Student-t, nu_i degrees of freedom, scale s_i: density only
  const T z = q / s;
  return -0.5 * (nu + 1.0) * log1p(z * z / nu);
}
"""


def test_coordinate_wise_density_only_form_matches_numpy(eng):
    """targets.Custom from the log-density alone (D = 70: register-resident kernel; D = 600: lock-step engine)"""
    from aehmc_amd import RandomStream, nuts, targets
    from test_gpu_custom_target import StudentT
    for D in (70, 600):
        r = np.random.default_rng(D)
        nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
        C, max_exp, eps = 3, 5, 0.35
        q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
        seeds = [40 + c for c in range(C)]
        tgt, otgt = targets.Custom(STUDENT_T_LOGP, params=[nu, s]), StudentT(nu, s)
        assert not tgt.hand_gradient
        kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
        state = nuts.new_state(dev(q0), tgt)
        for c in range(C):
            U, g = otgt(q0[c])
            np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
            np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-12, atol=1e-14)
        info, _ = kern(state, eps, imm)
        for c in range(C):
            o = no.nuts_kernel(no.RandomStream(seeds[c]), otgt, max_num_expansions=max_exp)(no.new_state(q0[c].copy(), otgt), eps, imm)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
            assert info.n_leapfrog[c].item() == o.n_leapfrog


LOGISTIC_LOGP = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) {
  return y * z - softplus(z);   // Bernoulli(logit = z)
}
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) {
  const double tau = prm[0][0];
  return -0.5 * q * q / (tau * tau);
}
"""


def test_glm_density_only_form_equals_the_hand_written_one(eng):
    """logistic regression from log-likelihood + log-prior alone == the hand-differentiated source of
    test_gpu_custom_target.py (same arithmetic up to the form of the derivative: 1e-10)"""
    from aehmc_amd import RandomStream, nuts, targets
    from test_gpu_custom_target import LOGISTIC
    r = np.random.default_rng(5)
    N, D, C = 300, 6, 4
    X = r.normal(size=(N, D))
    y = (r.random(N) < 1 / (1 + np.exp(-X @ r.normal(size=D)))).astype(np.float64)
    q0 = 0.3 * r.normal(size=(C, D))
    out = []
    for src in (LOGISTIC_LOGP, LOGISTIC):
        tgt = targets.CustomGLM(src, dev(X), dev(y), params=[[2.0]])
        kern = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=5)
        state = nuts.new_state(dev(q0), tgt)
        info, _ = kern(state, 0.05, np.ones(D))
        out.append((state.potential_energy.cpu().numpy(), state.potential_energy_grad.cpu().numpy(),
                    info.state.position.cpu().numpy(), info.n_leapfrog.cpu().numpy()))
    for a, b in zip(out[0][:3], out[1][:3]):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
    assert np.array_equal(out[0][3], out[1][3])


def test_a_wrong_hand_written_gradient_is_rejected(eng):
    """The reference cannot sample with a wrong gradient (it differentiates logprob_fn itself).  A hand-written
    aehmc_custom_elem / aehmc_glm_row whose gradient is not the derivative of its potential fails at new_state."""
    from aehmc_amd import nuts, targets
    from test_gpu_custom_target import STUDENT_T, LOGISTIC
    r = np.random.default_rng(8)
    D = 12
    nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
    q0 = dev(r.normal(size=(3, D)))
    good = targets.Custom(STUDENT_T, params=[nu, s])
    assert good.hand_gradient and not good.gradient_checked
    nuts.new_state(q0, good)
    assert good.gradient_checked
    bad = targets.Custom(STUDENT_T.replace("(nu + 1.0) * z / (nu + z * z) / s", "(nu + 1.0) * z / (nu + z * z)"), params=[nu, s])
    with pytest.raises(ValueError, match="hand-written gradient disagrees"):
        nuts.new_state(q0, bad)
    X, y = dev(r.normal(size=(50, D))), dev((r.random(50) < 0.5).astype(np.float64))
    nuts.new_state(q0, targets.CustomGLM(LOGISTIC, X, y, params=[[2.0]]))
    with pytest.raises(ValueError, match="hand-written gradient disagrees"):
        nuts.new_state(q0, targets.CustomGLM(LOGISTIC.replace("d = 1.0 / (1.0 + exp(-z)) - y;", "d = 1.0 / (1.0 + exp(-z));"),
                                            X, y, params=[[2.0]]))


def test_joint_density_errors(eng):
    from aehmc_amd import nuts, targets
    from aehmc_amd.engine import EngineError
    with pytest.raises(ValueError, match="dim <= 2048"):
        targets.CustomJoint(FUNNEL, dim=2049)
    with pytest.raises(EngineError, match="compilation failed"):
        nuts.new_state(dev(np.zeros((2, 4))), targets.CustomJoint(FUNNEL.replace("exp(-v)", "exq(-v)"), dim=4))
    # the engine is still bound to nothing broken: a good target right behind the failed one works
    st = nuts.new_state(dev(np.zeros((2, 4))), targets.CustomJoint(FUNNEL, dim=4))
    assert torch.isfinite(st.potential_energy).all()


@pytest.mark.parametrize("D", [65, 100, 130, 2048])
def test_joint_density_above_64_coordinates_value_and_gradient(eng, D):
    """new_state of a joint density with more coordinates than a wavefront has lanes: the row in LDS, ceil(D / 64) forward
    passes (k_target_joint_rows), ragged last pass; against the analytic gradient"""
    from aehmc_amd import nuts, targets
    r = np.random.default_rng(D)
    q0 = 0.3 * r.normal(size=(9, D))
    q0[:, 0] = 0.5 * r.normal(size=9)
    state = nuts.new_state(dev(q0), targets.CustomJoint(FUNNEL, dim=D))
    otgt = Funnel(D)
    for c in range(q0.shape[0]):
        U, g = otgt(q0[c])
        np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
        np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("D,metric,sampler", [(100, "diag", "nuts"), (100, "dense", "nuts"), (200, "diag", "nuts"),
                                              (100, "diag", "hmc"), (130, "dense", "hmc")])
def test_joint_density_above_64_coordinates_samplers_match_numpy(eng, D, metric, sampler):
    """NUTS / HMC with a joint density of 64 < D <= 2048 coordinates (lock-step path, the density evaluated between the
    stage kernels): every transition against the numpy restatement with the analytic gradient"""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(D + len(metric))
    C, n, max_exp, L = 4, 3, 5, 6
    eps = 0.05
    q0 = 0.3 * r.normal(size=(C, D))
    imm = make_metric(metric, D, r)
    seeds = [300 + c for c in range(C)]
    tgt, otgt = targets.CustomJoint(FUNNEL, dim=D), Funnel(D)
    mod, omod = (nuts, no.nuts_kernel) if sampler == "nuts" else (hmc, no.hmc_kernel)
    kw = {"max_num_expansions": max_exp} if sampler == "nuts" else {}
    kern = mod.new_kernel(RandomStream(seeds=seeds), tgt, **kw)
    state = mod.new_state(dev(q0), tgt)
    okern = [omod(no.RandomStream(sd), otgt, **kw) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    extra = () if sampler == "nuts" else (L,)
    for _ in range(n):
        info, _ = kern(state, eps, dev(imm) if metric == "dense" else imm, *extra)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, imm, *extra)
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-10)
            np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)
            assert bool(info.is_diverging[c]) == bool(o.is_diverging)
            if sampler == "nuts":
                assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
                assert bool(info.is_turning[c]) == bool(o.is_turning)


@pytest.mark.parametrize("D", [100, 192, 200])
def test_joint_density_above_64_one_launch_equals_lockstep_bitwise(eng, D):
    """64 < D, scalar / diagonal metric: k_nuts_joint_rows / k_hmc_joint_rows run the lock-step engine's own device
    functions for one chain per wavefront, all transitions of a sample() call in one launch -- bit for bit the lock-step
    path (resident_nuts / fused_hmc = 0: three launches per leapfrog), generator states included.  (NUTS takes the
    one-launch kernel up to D = 192, HMC at any size: D = 200 compares the lock-step NUTS with itself.)"""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(D)
    C = 7
    q0, imm = 0.3 * r.normal(size=(C, D)), 0.5 + r.random(D)
    outs = {}
    for fast in (1, 0):
        eng.set_option("resident_nuts", 2 if fast else 0)
        eng.set_option("fused_hmc", fast)
        tgt = targets.CustomJoint(AR1_CHAIN, dim=D)
        kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=5)
        sn, infn = kn.sample(nuts.new_state(dev(q0), tgt), 0.2, imm, 3)[:2]
        kh = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        sh, infh, acch = kh.sample(hmc.new_state(dev(q0), tgt), 0.2, imm, 5, 3)[:3]
        outs[fast] = (sn, infn.n_leapfrog, infn.state.potential_energy, sh, acch, kn._nuts["holder"]["rng"].clone(),
                      kh._hmc["holder"]["rng"].clone())
    for x, y in zip(outs[1], outs[0]):
        assert torch.equal(x, y)


def test_joint_density_above_64_per_chain_parameters(eng):
    """per-chain step sizes and per-chain diagonal metrics (what window adaptation hands back) on the one-launch joint
    kernels == the lock-step path, bit for bit; a per-chain DENSE metric (lock-step only) runs and stays finite"""
    from aehmc_amd import PerChain, RandomStream, hmc, nuts, targets
    r = np.random.default_rng(8)
    D, C = 90, 6
    q0 = 0.3 * r.normal(size=(C, D))
    eps = PerChain(dev(0.1 + 0.1 * r.random(C)))
    imm = PerChain(dev(0.5 + r.random((C, D))))
    outs = {}
    for fast in (1, 0):
        eng.set_option("resident_nuts", 2 if fast else 0)
        eng.set_option("fused_hmc", fast)
        tgt = targets.CustomJoint(AR1_CHAIN, dim=D)
        kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=5)
        sn, infn = kn.sample(nuts.new_state(dev(q0), tgt), eps, imm, 3)[:2]
        kh = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        sh = kh.sample(hmc.new_state(dev(q0), tgt), eps, imm, 4, 3)[0]
        outs[fast] = (sn, infn.n_leapfrog, sh)
    for x, y in zip(outs[1], outs[0]):
        assert torch.equal(x, y)
    A = r.normal(size=(C, D, D))
    dense = PerChain(dev(np.einsum("cij,ckj->cik", A, A) / D + np.eye(D)))
    tgt = targets.CustomJoint(AR1_CHAIN, dim=D)
    kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=4)
    info, _ = kn(nuts.new_state(dev(q0), tgt), 0.1, dense)
    assert torch.isfinite(info.state.position).all() and (info.n_leapfrog > 0).all()


def test_joint_density_on_the_lockstep_path_equals_the_single_launch_kernels(eng):
    """D <= 64 with resident_nuts / fused_hmc = 0: the same density on the lock-step path (it raised until round 5) --
    same trees, same accept decisions, values to rounding"""
    from aehmc_amd import RandomStream, hmc, nuts
    make, _, D = models()["schools"]
    r = np.random.default_rng(5)
    C = 6
    q0, imm = 0.5 * r.normal(size=(C, D)), 0.5 + r.random(D)
    outs = {}
    for fast in (1, 0):
        eng.set_option("resident_nuts", 2 if fast else 0)
        eng.set_option("fused_hmc", fast)
        tgt = make()
        kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        sn, infn = kn.sample(nuts.new_state(dev(q0), tgt), 0.1, imm, 4)[:2]
        kh = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        sh, infh = kh.sample(hmc.new_state(dev(q0), tgt), 0.1, imm, 7, 4)[:2]
        outs[fast] = (sn, infn.n_leapfrog, sh, infh.acceptance_probability)
    assert torch.equal(outs[1][1], outs[0][1])
    for k in (0, 2, 3):
        np.testing.assert_allclose(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy(), rtol=1e-9, atol=1e-12)


CORRELATED_NORMAL = """
// the reference's statistical test target (tests/test_hmc.py:190-264): 2-D normal, scales [1, 2], correlation 0.5
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  const double rho = 0.5, s0 = 1.0, s1 = 2.0;
  auto a = (q[0] - 1.0) / s0;
  auto b = (q[1] - 2.0) / s1;
  return -0.5 * (a * a - 2.0 * rho * a * b + b * b) / (1.0 - rho * rho);
}
"""


def test_joint_density_statistics_and_warmup(eng):
    """/root/reference/tests/test_hmc.py:190-264: HMC on a correlated 2-D normal defined by its joint density; mean,
    variances and correlation within Monte-Carlo error over many chains; window_adaptation.run takes the target"""
    from aehmc_amd import RandomStream, hmc, nuts, targets, window_adaptation
    tgt = targets.CustomJoint(CORRELATED_NORMAL, dim=2)
    C, n = 1024, 300
    r = np.random.default_rng(0)
    q0 = dev(r.normal(size=(C, 2)))
    kern = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    st = hmc.new_state(q0, tgt)
    burn = kern.sample(st, 0.25, np.array([1.0, 4.0]), 8, 100, keep_samples=False)[1]
    samples = kern.sample(burn.state._replace(momentum=None), 0.25, np.array([1.0, 4.0]), 8, n)[0].cpu().numpy()
    x = samples.reshape(-1, 2)
    per_chain_mean = samples.mean(axis=0)
    se = per_chain_mean.std(axis=0) / np.sqrt(C)
    assert np.all(np.abs(per_chain_mean.mean(axis=0) - [1.0, 2.0]) < 5 * se)
    assert abs(x[:, 0].var() - 1.0) < 0.05 and abs(x[:, 1].var() - 4.0) < 0.2
    assert abs(np.corrcoef(x.T)[0, 1] - 0.5) < 0.03
    # NUTS + window adaptation (diagonal) on the funnel: runs, adapts a finite step size and metric
    ftgt = targets.CustomJoint(FUNNEL, dim=6)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(64))), ftgt, max_num_expansions=6)
    state = nuts.new_state(dev(0.1 * r.normal(size=(64, 6))), ftgt)
    state, (eps, imm), _ = window_adaptation.run(kernel, state, num_steps=150)
    e = eps.value if hasattr(eps, "value") else eps
    assert torch.isfinite(torch.as_tensor(e)).all() and torch.isfinite(state.position).all()


AR1_CHAIN = """
// a stationary AR(1) chain x_0 ~ N(0, 1), x_i | x_{i-1} ~ N(rho x_{i-1}, 1 - rho^2): every marginal is N(0, 1), neighbours
// correlate with rho -- non-separable, any number of coordinates
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  const double rho = 0.6, s2 = 1.0 - rho * rho;
  auto lp = -0.5 * q[0] * q[0];
  for (int i = 1; i < q.size(); i++) {
    auto d = q[i] - rho * q[i - 1];
    lp += -0.5 * d * d / s2;
  }
  return lp;
}
"""


def test_joint_density_above_64_coordinates_statistics_and_warmup(eng):
    """An 80-coordinate non-separable density on the lock-step path under window_adaptation.run (per-chain step sizes and
    diagonal metrics) and NUTS sampling: unit marginal variances, lag-1 correlation rho, zero means within Monte-Carlo
    error -- as /root/reference/tests/test_hmc.py:267-346 checks its targets."""
    from aehmc_amd import RandomStream, nuts, targets, window_adaptation
    D, C, n = 80, 256, 120
    tgt = targets.CustomJoint(AR1_CHAIN, dim=D)
    r = np.random.default_rng(1)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    state = nuts.new_state(dev(r.normal(size=(C, D))), tgt)
    state, (eps, imm), _ = window_adaptation.run(kernel, state, num_steps=200)
    samples = kernel.sample(state, eps, imm, n)[0].cpu().numpy()  # [n, C, D]
    x = samples.reshape(-1, D)
    chain_means = samples.mean(axis=0)                              # [C, D]
    se = chain_means.std(axis=0) / np.sqrt(C)
    assert np.all(np.abs(chain_means.mean(axis=0)) < 5 * se + 1e-3)
    var = x.var(axis=0)
    assert np.all(np.abs(var - 1.0) < 0.12), (var.min(), var.max())
    lag1 = np.mean([np.corrcoef(x[:, i], x[:, i + 1])[0, 1] for i in range(0, D - 1, 7)])
    assert abs(lag1 - 0.6) < 0.03, lag1


def test_a_failed_compile_leaves_the_previous_target_bound(eng):
    """Binding a user-defined target is a transaction: a source that does not compile must not replace the
    code objects / parameter table of the target that is bound -- the next step of THAT target has to run on its own
    kernels and parameters, for a custom target and for a built-in one alike."""
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import EngineError
    r = np.random.default_rng(21)
    D, C = 8, 4
    nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
    q0, imm = dev(r.normal(size=(C, D))), 0.5 + r.random(D)
    good = targets.Custom(STUDENT_T_LOGP, params=[nu, s])
    bad = targets.Custom(STUDENT_T_LOGP.replace("log1p", "log1q"), params=[nu, s])
    builtin = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
    for tgt in (good, builtin):
        seeds = list(range(C))
        kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=5)
        state = nuts.new_state(q0, tgt)
        ref, _ = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=5)(state, 0.3, imm)
        with pytest.raises(EngineError, match="compilation failed"):
            nuts.new_state(q0, bad)
        info, _ = kern(state, 0.3, imm)  # the SAME target again: cache-key early return in Engine.set_target
        assert torch.equal(info.state.position, ref.state.position) and torch.equal(info.n_leapfrog, ref.n_leapfrog)


def test_compiled_targets_are_kept_on_disk_across_processes(tmp_path):
    """The code objects of a user-defined target are cached on disk (keyed by everything the compiler saw, in a directory
    named by the library's source hash): a second PROCESS binds the same target without recompiling and computes the
    same bits.  "Without recompiling" is the library's own count of hipRTC compilations (aehmc_rtc_stats), not a time."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
from test_gpu_autodiff import FUNNEL
torch.zeros(1, device="cuda")
tgt = targets.CustomJoint(FUNNEL, dim=10)
q0 = torch.as_tensor(0.3 * np.random.default_rng(0).normal(size=(8, 10)), device="cuda")
state = nuts.new_state(q0, tgt)
info, _ = nuts.new_kernel(RandomStream(seeds=list(range(8))), tgt, max_num_expansions=5)(state, 0.1, np.ones(10))
torch.cuda.synchronize()
print(json.dumps({"rtc": list(get_engine().rtc_stats()), "q": info.state.position.cpu().numpy().tolist()}))
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, AEHMC_AMD_RTC_CACHE=str(tmp_path))
    runs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    files = [f for d, _, fs in os.walk(tmp_path) for f in fs]
    assert len(files) == 2 and all(f.endswith(".aehmcco") for f in files), files  # new_state's program and the NUTS kernel
    assert runs[0]["q"] == runs[1]["q"]
    assert runs[0]["rtc"] == [2, 0] and runs[1]["rtc"] == [0, 2], [r["rtc"] for r in runs]  # [compiled, loaded from disk]
