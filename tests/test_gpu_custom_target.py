"""User-defined coordinate-wise targets (``targets.Custom``): the reference takes ANY logprob callable and
differentiates it (/root/reference/aehmc/hmc.py:16-40, integrators.py:61-65); here the user supplies the potential's
coordinate term and its derivative as HIP source and the engine compiles its kernel templates against it with hipRTC.
Parity: the same expression as a numpy callable through the numpy restatement (oracle/np_oracle.py), chain by chain on
identical seeds, 1e-9 with every discrete output identical; statistics as /root/reference/tests/test_hmc.py:267-346."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import np_oracle as no  # noqa: E402

RTOL = 1e-9

STUDENT_T = """
__device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g) {
  const double nu = prm[0][i], s = prm[1][i];   // Student-t, nu_i degrees of freedom, scale s_i
  const double z = q / s;
  u = 0.5 * (nu + 1.0) * log1p(z * z / nu);
  g = (nu + 1.0) * z / (nu + z * z) / s;
}
"""


class StudentT:
    """the numpy side of STUDENT_T: U = sum 0.5 (nu + 1) log1p(z^2 / nu), z = q / s"""

    def __init__(self, nu, s):
        self.nu, self.s = np.asarray(nu, dtype=np.float64), np.asarray(s, dtype=np.float64)

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        z = q / self.s
        u = 0.0
        for t in 0.5 * (self.nu + 1.0) * np.log1p(z * z / self.nu):  # sequential sum, as the restatement's targets
            u += t
        return float(u), (self.nu + 1.0) * z / (self.nu + z * z) / self.s


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


def oracle_nuts(tgt, seeds, q0, eps, imm, max_exp, n):
    out = []
    for c, seed in enumerate(seeds):
        kern = no.nuts_kernel(no.RandomStream(seed), tgt, max_num_expansions=max_exp)
        state, infos = no.new_state(q0[c].copy(), tgt), []
        for _ in range(n):
            info = kern(state, eps, imm)
            infos.append(info)
            state = info.state._replace(momentum=None)
        out.append(infos)
    return out


@pytest.mark.parametrize("D,metric,resident", [(10, "diag", 2), (10, "diag", 0), (70, "diag", 2), (300, "diag", 2),
                                               (40, "dense", 2), (90, "dense", 2), (200, "dense", 2), (300, "dense", 2),
                                               (600, "diag", 2), (600, "diag", 0),
                                               (1500, "diag", 2), (5000, "diag", 2), (10176, "diag", 2)])
def test_custom_target_nuts_matches_numpy(eng, D, metric, resident):
    """register-resident kernel (D <= 512, diagonal / scalar metric), workgroup-per-chain kernel (512 < D <= 10176 -- the
    largest chain whose q and dU/dq fit the CU's 160 KB of LDS, tested at exactly that size:
    k_nuts_wide, q and dU/dq in LDS above D = 4096; round 5), block-resident kernels (shared dense metric, 64 < D <= 512:
    k_nuts_block_reg / k_nuts_block_dense; round 5), lock-step engine (resident_nuts = 0, small dense problems): all
    compiled at run time against the user's function"""
    from aehmc_amd import RandomStream, nuts, targets
    eng.set_option("resident_nuts", resident)
    r = np.random.default_rng(D + len(metric))
    nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
    C, n, max_exp, eps = 4, 3, 5, 0.35
    q0 = r.normal(size=(C, D))
    if metric == "scalar":
        imm = np.float64(0.8)
    elif metric == "diag":
        imm = 0.5 + r.random(D)
    else:
        A = r.normal(size=(D, D))
        imm = A @ A.T / D + np.eye(D)
        imm = 0.5 * (imm + imm.T)
    seeds = [40 + c for c in range(C)]
    tgt = targets.Custom(STUDENT_T, params=[nu, s])
    kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    ref = oracle_nuts(StudentT(nu, s), seeds, q0, eps, imm, max_exp, n)
    np.testing.assert_allclose(state.potential_energy.cpu().numpy(), [StudentT(nu, s)(q0[c])[0] for c in range(C)], rtol=1e-12)
    for t in range(n):
        info, _ = kern(state, eps, dev(imm) if metric == "dense" else imm)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = ref[c][t]
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)


@pytest.mark.parametrize("D,fused,metric", [(10, 1, "diag"), (10, 0, "diag"), (200, 1, "diag"), (1500, 1, "diag"),
                                            (3000, 1, "diag"), (5000, 1, "diag"), (9000, 1, "diag"), (10240, 1, "diag"),
                                            (100, 1, "dense"), (200, 1, "dense"), (300, 1, "dense"), (200, 0, "dense")])
def test_custom_target_hmc_matches_numpy(eng, D, fused, metric):
    """fused register-resident HMC (D <= 1024), the workgroup-per-chain kernel (1024 < D <= 10240: k_hmc_wide at 256 /
    512 / 1024 threads; round 5), the block-resident kernels (shared dense metric, 64 < D <= 512: k_hmc_block_reg /
    k_hmc_block_dense; round 5) and the lock-step engine, all compiled at run time against the user's function"""
    from aehmc_amd import RandomStream, hmc, targets
    eng.set_option("fused_hmc", fused)
    r = np.random.default_rng(D)
    nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
    C, n, L, eps = 3, 3, 7, 0.2
    q0 = r.normal(size=(C, D))
    if metric == "scalar":
        imm = np.float64(0.8)
    elif metric == "diag":
        imm = 0.5 + r.random(D)
    else:
        A = r.normal(size=(D, D))
        imm = A @ A.T / D + np.eye(D)
        imm = 0.5 * (imm + imm.T)
    seeds = [90 + c for c in range(C)]
    tgt, otgt = targets.Custom(STUDENT_T, params=[nu, s]), StudentT(nu, s)
    kern = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    state = hmc.new_state(dev(q0), tgt)
    okern = [no.hmc_kernel(no.RandomStream(sd), otgt) for sd in seeds]
    ostate = [no.new_state(q0[c].copy(), otgt) for c in range(C)]
    for _ in range(n):
        info, _ = kern(state, eps, dev(imm) if metric == "dense" else imm, L)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = okern[c](ostate[c], eps, imm, L)
            ostate[c] = o.state._replace(momentum=None)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad,
                                       rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)
            assert bool(info.is_diverging[c]) == bool(o.is_diverging)


@pytest.mark.parametrize("D,metric", [(1500, "diag"), (200, "dense")])
def test_custom_target_wide_and_block_hmc_sample_equals_lockstep(eng, D, metric):
    """sample(T) of a user-defined target on k_hmc_wide / k_hmc_block_reg against the lock-step engine with the same user
    function: same accept decisions and generator states, positions to rounding (the cross-wavefront sums of the
    workgroup-per-chain kernel are ordered differently) / bit for bit (block kernels)"""
    from aehmc_amd import RandomStream, hmc, targets
    r = np.random.default_rng(D + 1)
    C = 5
    nu, s = 4.0 + r.random(D), 0.7 + r.random(D)
    q0 = r.normal(size=(C, D))
    if metric == "diag":
        imm = 0.5 + r.random(D)
    else:
        A = r.normal(size=(D, D))
        imm = dev(A @ A.T / D + np.eye(D))
    tgt = targets.Custom(STUDENT_T, params=[dev(nu), dev(s)])
    outs = []
    for fast in (1, 0):
        eng.set_option("fused_hmc", fast)
        kern = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        res = kern.sample(hmc.new_state(dev(q0), tgt), 0.15, imm, 6, 5)
        outs.append((res[0], res[2], kern._hmc["holder"]["rng"].cpu().numpy().copy()))
    (s1, a1, g1), (s0, a0, g0) = outs
    assert np.array_equal(g1, g0)
    if metric == "dense":
        assert torch.equal(s1, s0) and torch.equal(a1, a0)
    else:
        np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=1e-9)


def test_custom_target_fused_paths_equal_lockstep_bitwise(eng):
    """sample(T) on the run-time compiled fused kernels == the lock-step engine with the same user function, bit for bit
    (D = 100: one wavefront per chain in both)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(2)
    D, C = 100, 6
    nu, s = 4.0 + r.random(D), 0.7 + r.random(D)
    q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
    tgt = targets.Custom(STUDENT_T, params=[dev(nu), dev(s)])

    def run(mod, fast, *extra):
        eng.set_option("resident_nuts", 2 if fast else 0)
        eng.set_option("fused_hmc", 1 if fast else 0)
        eng.set_option("resident_min_team", 0)
        kern = mod.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        out = kern.sample(mod.new_state(dev(q0), tgt), 0.3, imm, *extra, 4)
        return out[0], out[2]

    for mod, extra in ((hmc, (9,)), (nuts, ())):
        s1, a1 = run(mod, True, *extra)
        s0, a0 = run(mod, False, *extra)
        if mod is hmc:
            assert torch.equal(s1, s0) and torch.equal(a1, a0)
        else:  # C = 6 chains of D = 100: a 64-lane team per chain, the lock-step path's summation order
            assert torch.equal(s1, s0) and torch.equal(a1, a0)


def test_custom_target_statistics_and_warmup(eng):
    """window adaptation + sampling of a user-defined target (the loop of tests/test_hmc.py:190-346): Student-t(5)
    coordinates with scales 1 and 3 -- mean 0, variance s^2 nu / (nu - 2).  HMC's moments are held to Monte-Carlo
    error; NUTS reproduces the reference's upward variance bias (its 2**j + 1 leapfrogs per sub-trajectory:
    tests/test_nuts_quirks.py), so its variance is only bracketed."""
    from aehmc_amd import RandomStream, hmc, nuts, targets, window_adaptation
    C, D = 2048, 2
    nu, s = np.array([5.0, 5.0]), np.array([1.0, 3.0])
    var_true = s ** 2 * nu / (nu - 2)
    tgt = targets.Custom(STUDENT_T, params=[nu, s])
    q0 = dev(np.random.default_rng(0).normal(size=(C, D)))
    for mod in (hmc, nuts):
        kern = mod.new_kernel(RandomStream(seeds=list(range(7000, 7000 + C))), tgt)
        state = mod.new_state(q0, tgt)
        extra = dict(num_integration_steps=7) if mod is hmc else {}
        state, (eps, imm), _ = window_adaptation.run(kern, state, 300, **extra)
        out = kern.sample(state, eps, imm, *((7,) if mod is hmc else ()), 400)
        x, acc = out[0].cpu().numpy(), out[2]  # [400, C, 2]
        mean, var = x.mean(axis=(0, 1)), x.var(axis=(0, 1))
        # chains are independent: Monte-Carlo error from the spread of the per-chain statistics
        se_mean = x.mean(axis=0).std(axis=0) / np.sqrt(C)
        se_var = x.var(axis=0).std(axis=0) / np.sqrt(C)
        assert np.all(np.abs(mean) < 5 * se_mean + 1e-3), (mod.__name__, mean, se_mean)
        if mod is hmc:
            assert np.all(np.abs(var - var_true) < 5 * se_var + 0.02 * var_true), (var, var_true, se_var)
        else:
            assert np.all(var > 0.98 * var_true) and np.all(var < 1.2 * var_true), (var, var_true)
        assert 0.6 < float(acc.mean()) < 0.97


def test_custom_target_compile_error_is_reported(eng):
    from aehmc_amd import nuts, targets
    from aehmc_amd.engine import EngineError
    bad = targets.Custom("__device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, "
                         "double &g) { u = undefined_symbol(q); g = q; }")
    with pytest.raises(EngineError, match="compilation failed"):
        nuts.new_state(dev(np.zeros((2, 3))), bad)
    # the engine is usable afterwards
    ok = targets.Custom(STUDENT_T, params=[np.full(3, 4.0), np.ones(3)])
    st = nuts.new_state(dev(np.ones((2, 3))), ok)
    np.testing.assert_allclose(st.potential_energy.cpu().numpy(), [StudentT(np.full(3, 4.0), np.ones(3))(np.ones(3))[0]] * 2, rtol=1e-12)


def test_custom_target_fp_contract_hmc(eng):
    """the fast-arithmetic mode of the fused HMC kernel with a user-defined target: within 1e-6 of the default mode"""
    from aehmc_amd import RandomStream, hmc, targets
    for D in (64, 1500):  # k_hmc_fused, k_hmc_wide (round 5)
        r = np.random.default_rng(3)
        C = 5
        nu, s = 4.0 + r.random(D), 0.7 + r.random(D)
        q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
        tgt = targets.Custom(STUDENT_T, params=[nu, s])
        outs = []
        for fc in (0, 1):
            eng.set_option("fp_contract", fc)
            kern = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            outs.append(kern.sample(hmc.new_state(dev(q0), tgt), 0.1 if D == 64 else 0.05, imm, 20, 3)[0])
        assert not torch.equal(outs[0], outs[1])
        np.testing.assert_allclose(outs[1].cpu().numpy(), outs[0].cpu().numpy(), rtol=1e-6, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------
# row-reduction ("GLM-type") user-defined targets: logistic regression
LOGISTIC = """
__device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &l, double &d) {
  l = (z > 0 ? z + log1p(exp(-z)) : log1p(exp(z))) - y * z;   // log(1 + e^z) - y z
  d = 1.0 / (1.0 + exp(-z)) - y;
}
__device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g) {
  const double tau = prm[0][0];                                  // N(0, tau^2) prior on every weight
  u = 0.5 * q * q / (tau * tau);
  g = q / (tau * tau);
}
"""


class Logistic:
    """the numpy side of LOGISTIC"""

    def __init__(self, X, y, tau):
        self.X, self.y, self.tau = np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64), float(tau)

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        z = self.X @ q
        loss = np.where(z > 0, z + np.log1p(np.exp(-np.abs(z))), np.log1p(np.exp(-np.abs(z)))) - self.y * z
        d = 1.0 / (1.0 + np.exp(-z)) - self.y
        U = float(loss.sum() + (0.5 * q * q / self.tau ** 2).sum())
        return U, self.X.T @ d + q / self.tau ** 2


def logistic_data(N, D, seed):
    r = np.random.default_rng(seed)
    X = r.normal(size=(N, D))
    w = r.normal(size=D)
    y = (r.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
    return X, y, w


@pytest.mark.parametrize("N,D,metric,resident", [(300, 6, "diag", 2), (300, 6, "diag", 0), (1000, 3, "scalar1", 2), (257, 20, "dense", 2),
                                                 (64, 70, "dense", 2), (2000, 8, "diag", 2), (700, 13, "diag", 2), (130, 32, "diag", 2),
                                                 (500, 33, "diag", 2)])
def test_custom_glm_target_matches_numpy(eng, N, D, metric, resident):
    """logistic regression through NUTS and HMC against the numpy restatement, chain by chain: on the lock-step engine
    (the two products with X on the fp64 MFMA GEMM, the user's loss / prior in run-time compiled kernels: dense metrics,
    D > 32, resident_nuts = 0) and, round 5, in ONE launch per call for D <= 32 with a scalar / diagonal metric (k_nuts_glm_rows
    / k_hmc_glm_rows<8|16|32>: the wavefront that owns a chain sweeps the data rows itself)"""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    eng.set_option("resident_nuts", resident)
    eng.set_option("fused_hmc", 1 if resident else 0)
    X, y, _ = logistic_data(N, D, N + D)
    r = np.random.default_rng(D)
    tau, C = 2.0, 4
    q0 = 0.3 * r.normal(size=(C, D))
    if metric == "dense":
        A = r.normal(size=(D, D))
        imm = 0.05 * (A @ A.T / D + np.eye(D))
        imm = 0.5 * (imm + imm.T)
    elif metric == "diag":
        imm = 0.02 + 0.05 * r.random(D)
    else:
        imm = np.full(D, 0.01)
    immg = dev(imm) if metric == "dense" else imm
    tgt, otgt = targets.CustomGLM(LOGISTIC, X, y, params=[[tau]]), Logistic(X, y, tau)
    seeds = [11 + c for c in range(C)]
    state = nuts.new_state(dev(q0), tgt)
    for c in range(C):
        U, g = otgt(q0[c])
        np.testing.assert_allclose(state.potential_energy[c].item(), U, rtol=1e-12)
        np.testing.assert_allclose(state.potential_energy_grad[c].cpu().numpy(), g, rtol=1e-10, atol=1e-11)
    kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=5)
    ref = oracle_nuts(otgt, seeds, q0, 0.5, imm, 5, 2)
    for t in range(2):
        info, _ = kern(state, 0.5, immg)
        state = info.state._replace(momentum=None)
        for c in range(C):
            o = ref[c][t]
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
            assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
            assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)
    hk = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    hstate = hmc.new_state(dev(q0), tgt)
    info, _ = hk(hstate, 0.3, immg, 6)
    for c in range(C):
        o = no.hmc_kernel(no.RandomStream(seeds[c]), otgt)(no.new_state(q0[c].copy(), otgt), 0.3, imm, 6)
        np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
        np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)


def test_custom_glm_one_launch_equals_lockstep(eng):
    """sample(T) of a row-reduction target on the one-launch kernels against the lock-step engine (GEMMs): same trees, same
    accept decisions, same generator states; values to rounding (the row sums run in another order than the GEMM's)"""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    N, D, C = 900, 10, 9
    X, y, _ = logistic_data(N, D, 77)
    r = np.random.default_rng(4)
    q0, imm = 0.3 * r.normal(size=(C, D)), 0.02 + 0.05 * r.random(D)
    tgt = targets.CustomGLM(LOGISTIC, X, y, params=[[2.0]])
    outs = {}
    for fast in (1, 0):
        eng.set_option("resident_nuts", 2 if fast else 0)
        eng.set_option("fused_hmc", fast)
        kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        sn, infn = kn.sample(nuts.new_state(dev(q0), tgt), 0.4, imm, 4)[:2]
        kh = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        sh, _, acch = kh.sample(hmc.new_state(dev(q0), tgt), 0.3, imm, 6, 4)[:3]
        outs[fast] = (sn, sh, acch, infn.n_leapfrog, kn._nuts["holder"]["rng"].clone(), kh._hmc["holder"]["rng"].clone())
    for k in (3, 4, 5):
        assert torch.equal(outs[1][k], outs[0][k])
    for k in (0, 1, 2):
        np.testing.assert_allclose(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("N, D", [(9000, 5), (8193, 17)])
def test_custom_glm_workgroup_per_chain_equals_wavefront_per_chain(eng, N, D):
    """Long data, few chains: eight wavefronts share a chain's sweep over the rows (glm_rows.cuh: k_nuts_glm_wg /
    k_hmc_glm_wg, taken for N >= 8192 and <= 2048 chains; option joint_wg = 0 keeps a wavefront per chain).  Same trees,
    accept decisions and generator states, values to rounding (the row sums are associated differently)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    C = 5
    X, y, _ = logistic_data(N, D, 31)
    r = np.random.default_rng(6)
    q0, imm = 0.2 * r.normal(size=(C, D)), 0.002 + 0.005 * r.random(D)
    tgt = targets.CustomGLM(LOGISTIC, dev(X), dev(y), params=[[2.0]])
    outs = {}
    try:
        for wg in (1, 0):
            eng.set_option("joint_wg", wg)
            kn = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
            sn, infn = kn.sample(nuts.new_state(dev(q0), tgt), 0.4, imm, 4)[:2]
            kh = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            sh, _, acch = kh.sample(hmc.new_state(dev(q0), tgt), 0.3, imm, 6, 4)[:3]
            outs[wg] = (sn, sh, acch, infn.n_leapfrog, kn._nuts["holder"]["rng"].clone(), kh._hmc["holder"]["rng"].clone())
    finally:
        eng.set_option("joint_wg", 1)
    assert int(outs[1][3].sum()) > 4 * C
    for k in (3, 4, 5):
        assert torch.equal(outs[1][k], outs[0][k])
    for k in (0, 1, 2):
        np.testing.assert_allclose(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("N, D, waves", [(63, 3, 0), (64, 8, 0), (65, 8, 3), (2500, 8, 4), (100_000, 8, 0), (9000, 21, 0)])
def test_custom_glm_workgroup_kernels_match_numpy(eng, N, D, waves):
    """The workgroup-per-chain kernels (forced: joint_wg = 2) against the numpy restatement, chain by chain, from one
    wavefront's worth of rows to 10^5; compiled for three / four wavefronts per SIMD (option wg_waves; 0 = the engine picks)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    X, y, w = logistic_data(N, D, N + D)
    r = np.random.default_rng(D)
    tau, C = 2.0, 3
    q0 = (w if N > 1000 else 0.0) + 0.3 * r.normal(size=(C, D)) / np.sqrt(N / 100.0)  # (long data: a start near the narrow posterior)
    imm = (0.02 + 0.05 * r.random(D)) * min(1.0, 300.0 / N)
    tgt, otgt = targets.CustomGLM(LOGISTIC, dev(X), dev(y), params=[[tau]]), Logistic(X, y, tau)
    seeds = [11 + c for c in range(C)]
    try:
        eng.set_option("joint_wg", 2)
        eng.set_option("wg_waves", waves)
        state = nuts.new_state(dev(q0), tgt)
        kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=5)
        ref = oracle_nuts(otgt, seeds, q0, 0.5, imm, 5, 2)
        nl = 0
        for t in range(2):
            info, _ = kern(state, 0.5, imm)
            state = info.state._replace(momentum=None)
            for c in range(C):
                o = ref[c][t]
                np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
                np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL)
                np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad, rtol=1e-8, atol=1e-8)
                assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings
                assert bool(info.is_turning[c]) == bool(o.is_turning) and bool(info.is_diverging[c]) == bool(o.is_diverging)
                nl += o.n_leapfrog
        assert nl > 2 * C
        hk = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
        info, _ = hk(hmc.new_state(dev(q0), tgt), 0.3, imm, 6)
        for c in range(C):
            o = no.hmc_kernel(no.RandomStream(seeds[c]), otgt)(no.new_state(q0[c].copy(), otgt), 0.3, imm, 6)
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11)
            np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-8)
    finally:
        eng.set_option("joint_wg", 1)
        eng.set_option("wg_waves", 0)


def test_custom_glm_posterior(eng):
    """Bayesian logistic regression sampled with NUTS after window adaptation (the use the reference's README and
    notebook show for its own models): the posterior mean recovers the generating weights within its own spread, and
    HMC with the adapted parameters gives the same posterior means within Monte-Carlo error."""
    from aehmc_amd import RandomStream, hmc, nuts, targets, window_adaptation
    N, D, C = 2000, 5, 512
    X, y, w = logistic_data(N, D, 5)
    tgt = targets.CustomGLM(LOGISTIC, dev(X), dev(y), params=[[3.0]])
    kern = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    state = nuts.new_state(dev(np.zeros((C, D))), tgt)
    state, (eps, imm), _ = window_adaptation.run(kern, state, 200)
    samples, info, acc, div = kern.sample(state, eps, imm, 150)
    x = samples.cpu().numpy()
    mean, sd = x.mean(axis=(0, 1)), x.std(axis=(0, 1))
    assert np.all(np.abs(mean - w) < 4 * sd), (mean, w, sd)   # N = 2000 rows: sd ~ 0.06
    assert np.all(sd < 0.2) and not bool(div.any()) and 0.6 < float(acc.mean()) < 0.97
    hk = hmc.new_kernel(RandomStream(seeds=list(range(C, 2 * C))), tgt)
    hs, _, hacc, _ = hk.sample(state, eps, imm, 8, 150)
    hmean = hs.cpu().numpy().mean(axis=(0, 1))
    se = x.mean(axis=0).std(axis=0) / np.sqrt(C)
    assert np.all(np.abs(hmean - mean) < 6 * se + 2e-3), (hmean, mean, se)
