"""Pin the CPU restatements (oracle/np_oracle.py, oracle/c/aehmc_oracle.c) against every
value the reference publishes for the HMC/NUTS path: README G1, notebook
G2/G3 and the known-answer tables of the reference's own unit tests."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no

G1_POSITION = 1.1034719409361107  # /root/reference/README.md:53-54


# ------------------------------------------------------------------ G1 (README.md:22-54)
def test_g1_numpy_bit_exact():
    srng = no.RandomStream(0)
    kernel = no.nuts_kernel(srng, no.StdNormal())
    state = no.new_state(np.float64(0.0), no.StdNormal())
    trace = []
    info = kernel(state, 1e-2, np.float64(1.0), trace=trace)
    assert float(info.state.position) == G1_POSITION
    assert info.num_doublings == 8 and not info.is_diverging and not info.is_turning
    assert [t["sub_len"] for t in trace] == [2, 3, 5, 9, 17, 33, 65, 2]  # 2**j+1 quirk
    assert info.n_leapfrog == 136


def test_g1_c_bit_exact():
    t = co.Target(co.T_STD_NORMAL, 1)
    m = co.Metric(np.float64(1.0), 1)
    rng = co.site_states([0], 4)
    q, U, g = co.new_state(t, [[0.0]])
    r = co.nuts_step(t, m, rng, 1e-2, q, U, g)
    assert q[0, 0] == G1_POSITION
    assert r["num_doublings"][0] == 8 and r["n_leapfrog"][0] == 136
    assert r["acceptance_probability"][0] == pytest.approx(0.9999767760191554, rel=1e-14)


# ------------------------------------------------------------------ G2/G3 (notebook)
def test_g3_regression_logprob(regression_data):
    X, y = regression_data
    lp = no.LinearRegression(X, y).logp(np.array([3.0, np.log(10.0)]))
    # examples/LinearRegression.ipynb:188
    assert lp == pytest.approx(-32238.026021294307, rel=1e-12)
    t = co.Target(co.T_LINREG, 2, X=X, y=y)
    _, U, _ = co.new_state(t, [[3.0, np.log(10.0)]])
    assert -U[0] == pytest.approx(-32238.026021294307, rel=1e-12)


def _check_g2(q, U, g, pacc, div):
    # examples/LinearRegression.ipynb:293-297 (printed to 8 decimals)
    np.testing.assert_allclose(q, [2.99946192, -1.30494977], atol=5e-9)
    assert U == pytest.approx(12433.00653542, abs=5e-9)
    np.testing.assert_allclose(g, [-489.93218536, -22571.36970197], atol=5e-9)
    assert pacc == 1.0 and not div


def test_g2_hmc_numpy(regression_data):
    X, y = regression_data
    target = no.LinearRegression(X, y)
    kernel = no.hmc_kernel(no.RandomStream(0), target)
    state = no.new_state(np.array([3.0, np.log(0.21)]), target)
    info = kernel(state, 5e-5, np.array([1.0, 1.0]), 1024)
    _check_g2(info.state.position, info.state.potential_energy,
              info.state.potential_energy_grad, info.acceptance_probability,
              info.is_diverging)


def test_g2_hmc_c(regression_data):
    X, y = regression_data
    t = co.Target(co.T_LINREG, 2, X=X, y=y)
    m = co.Metric(np.array([1.0, 1.0]), 2)
    rng = co.site_states([0], 2)
    q, U, g = co.new_state(t, [[3.0, np.log(0.21)]])
    r = co.hmc_step(t, m, rng, 5e-5, 1024, q, U, g)
    _check_g2(q[0], U[0], g[0], r["acceptance_probability"][0], r["is_diverging"][0])


# ------------------------------------------------------------------ tests/test_termination.py
CKPT = np.array([1.0, 2.0, 3.0, -2.0])
CKPT_SUM = np.array([2.0, 4.0, 4.0, -1.0])


@pytest.mark.parametrize("idx, expected", [((3, 3), True), ((3, 2), False), ((0, 0), False),
                                           ((0, 1), True), ((1, 3), True)])
def test_iterative_turning_table(idx, expected):
    # tests/test_termination.py:12-48
    _, _, is_turning, _ = no.gaussian_metric(np.float64(1.0))
    _, _, is_iter = no.iterative_uturn(is_turning)
    st = no.TerminationState(CKPT, CKPT_SUM, idx[0], idx[1])
    assert is_iter(st, np.float64(3.0), np.float64(1.0)) == expected
    for imm in (np.float64(1.0), np.ones(1)):
        m = co.Metric(imm, 1)
        assert co.is_iterative_turning(m, CKPT, CKPT_SUM, idx[0], idx[1], 3.0, 1.0) == expected


@pytest.mark.parametrize("step, expected", [(0, (1, 0)), (6, (3, 2)), (7, (0, 2)),
                                            (13, (2, 2)), (15, (0, 3))])
def test_find_storage_indices_table(step, expected):
    # tests/test_termination.py:51-62
    assert no.find_storage_indices(step) == expected
    assert co.find_storage_indices(step) == expected


def test_find_storage_indices_closed_form():
    """literal loops == (popcount(step>>1) - trailing_ones(step) + 1, popcount(step>>1))."""
    for step in range(1, 2048):
        n = 0
        while (step >> n) & 1:
            n += 1
        mx = bin(step >> 1).count("1")
        assert co.find_storage_indices(step) == (mx - n + 1, mx)
        assert no.find_storage_indices(step) == (mx - n + 1, mx)


@pytest.mark.parametrize("num_dims", [1, 3])
def test_termination_update_odd_step(num_dims):
    # tests/test_termination.py:65-90: an odd step leaves the checkpoints untouched
    _, _, is_turning, _ = no.gaussian_metric(np.ones(1))
    new_state, update, _ = no.iterative_uturn(is_turning)
    st = new_state(np.ones(num_dims), 4)
    st2 = update(st, np.ones(num_dims), np.ones(num_dims), 1)
    assert np.all(st2.momentum_checkpoints == 0) and np.all(st2.momentum_sum_checkpoints == 0)


# ------------------------------------------------------------------ tests/test_metrics.py
@pytest.mark.parametrize("imm, p, expected", [(np.float64(1.0), np.float64(1.0), 0.5),
                                              (np.ones(1), np.ones(1), 0.5),
                                              (np.ones(2), np.ones(2), 1.0),
                                              (np.eye(2), np.ones(2), 1.0)])
def test_kinetic_energy_table(imm, p, expected):
    # tests/test_metrics.py:39-68
    _, ke, _, _ = no.gaussian_metric(imm)
    assert ke(p) == expected and np.ndim(ke(p)) == 0
    assert co.kinetic_energy(co.Metric(imm, np.size(p)), p) == expected


@pytest.mark.parametrize("imm, n", [(np.float64(1.0), 1), (np.ones(2), 2), (np.eye(2), 2)])
def test_is_turning_rho_zero(imm, n):
    # tests/test_metrics.py:71-120: p_l = p_r = p_sum = ones -> rho = 0 -> `<= 0` -> True
    _, _, is_turning, _ = no.gaussian_metric(imm)
    p = np.ones(n) if n > 1 else np.float64(1.0)
    assert is_turning(p, p, p) is True
    assert co.is_turning(co.Metric(imm, n), np.ones(n), np.ones(n), np.ones(n)) is True


def test_mass_matrix_3d_raises():
    # tests/test_metrics.py:123-127
    with pytest.raises(ValueError):
        no.gaussian_metric(np.ones((2, 2, 2)))
    with pytest.raises(ValueError):
        co.Metric(np.ones((2, 2, 2)), 2)


# ------------------------------------------------------------------ tests/test_integrators.py
def test_velocity_verlet_harmonic_oscillator():
    # tests/test_integrators.py:58-67,101-131: 100 steps of 0.01 from (0, 1) -> (sin 1, cos 1)
    t = co.Target(co.T_ISO_GAUSSIAN, 1)
    m = co.Metric(np.ones(1), 1)
    q, U, g = co.new_state(t, [[0.0]])
    p = np.array([[1.0]])
    e0 = U[0] + 0.5
    co.leapfrog(t, m, 0.01, 100, q, p, U, g)
    assert q[0, 0] == pytest.approx(np.sin(1.0), abs=1e-2)
    assert p[0, 0] == pytest.approx(np.cos(1.0), abs=1e-2)
    assert U[0] + 0.5 * p[0, 0] ** 2 == pytest.approx(e0, rel=1e-4)
    # numpy restatement, same trajectory bit for bit
    _, _, _, vel = no.gaussian_metric(np.ones(1))
    step = no.velocity_verlet(no.IsoGaussian(), vel)
    s = no.new_state(np.array([0.0]), no.IsoGaussian())._replace(momentum=np.array([1.0]))
    for _ in range(100):
        s = step(s, 0.01)
    assert s.position[0] == q[0, 0] and s.momentum[0] == p[0, 0]


# ------------------------------------------------------------------ tests/test_trajectory.py
@pytest.mark.parametrize("step_size, div, turn, doublings",
                         [(100000.0, True, False, 1), (0.0000001, False, False, 10),
                          (1.0, False, True, 1)])
def test_multiplicative_expansion_outcomes(step_size, div, turn, doublings):
    # tests/test_trajectory.py:144-208: U = x^2/2, q = 1, imm = 1.0, seed 59, max 10 expansions
    kernel = no.nuts_kernel(no.RandomStream(59), no.IsoGaussian())
    info = kernel(no.new_state(np.float64(1.0), no.IsoGaussian()), step_size, np.float64(1.0))
    assert (info.is_diverging, info.is_turning, info.num_doublings) == (div, turn, doublings)
    if doublings == 10:
        assert info.n_leapfrog == 1033  # full tree: sum(2**j + 1)
    t = co.Target(co.T_ISO_GAUSSIAN, 1)
    q, U, g = co.new_state(t, [[1.0]])
    r = co.nuts_step(t, co.Metric(np.float64(1.0), 1), co.site_states([59], 4), step_size, q, U, g)
    assert (bool(r["is_diverging"][0]), bool(r["is_turning"][0]), r["num_doublings"][0]) == \
        (div, turn, doublings)
    assert q[0, 0] == float(info.state.position) and r["n_leapfrog"][0] == info.n_leapfrog


@pytest.mark.parametrize("step_size, div, term", [(1e-7, False, False), (1000.0, True, False),
                                                   (1e100, True, False)])
def test_dynamic_integration_outcomes(step_size, div, term):
    # tests/test_trajectory.py:77-141: N(0,1), q = ones(1), imm = ones(1), 10 steps
    target = no.StdNormal()
    mom, ke, is_turning, vel = no.gaussian_metric(np.ones(1))
    new_t, upd, crit = no.iterative_uturn(is_turning)
    srng = no.RandomStream(59)
    g_m, g_u = srng.site(), srng.site()
    integ = no.dynamic_integration(g_u, no.velocity_verlet(target, vel), ke, upd, crit, 1000)
    s = no.new_state(np.ones(1), target)._replace(momentum=mom(g_m))
    e0 = s.potential_energy + ke(s.momentum)
    with np.errstate(all="ignore"):
        out = integ(s, 1.0, new_t(s.position, 10), 10, step_size, e0)
    assert (out[5], out[6]) == (div, term)


# ------------------------------------------------------------------ numpy <-> C restatements
@pytest.mark.parametrize("kind", ["scalar", "diag", "dense", "dense40"])
def test_c_matches_numpy_nuts(kind):
    """Two independent restatements (numpy: the reference's own array expressions with numpy /
    scipy.linalg in place of Aesara ops; C: explicit loops) agree on whole transitions.  The dense
    branch has no reference-held value: this cross-check, at D = 5 and D = 40, is what
    stands behind it."""
    D = 1 if kind == "scalar" else 40 if kind == "dense40" else 5  # reference: ndim-0 imm goes with a scalar position
    kind = "dense" if kind == "dense40" else kind
    r = np.random.default_rng(3)
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    if kind == "dense":
        A = r.normal(size=(D, D))
        imm = A @ A.T / D + np.eye(D)
        prec = np.linalg.inv(imm)
        nt, ct_ = no.DenseMVN(mu, prec), co.Target(co.T_DENSE_MVN, D, mu=mu, prec=prec)
    else:
        imm = np.float64(0.7) if kind == "scalar" else 0.5 + r.random(D)
        nt, ct_ = no.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    m = co.Metric(imm, D)
    for seed in (11, 12, 13):
        q0 = r.normal(size=D)
        kernel = no.nuts_kernel(no.RandomStream(seed), nt, max_num_expansions=6)
        if kind == "scalar":
            nt.mu, nt.sigma = nt.mu[:1], nt.sigma[:1]
        rng = co.site_states([seed], 4)
        q, U, g = co.new_state(ct_, q0[None])
        st = no.new_state(q0[0] if kind == "scalar" else q0, nt)
        for _ in range(3):
            info = kernel(st, 0.3, imm)
            res = co.nuts_step(ct_, m, rng, 0.3, q, U, g, max_exp=6)
            st = info.state._replace(momentum=None)
            np.testing.assert_allclose(q[0], info.state.position, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(res["momentum"][0], info.state.momentum, rtol=1e-10, atol=1e-12)
            assert U[0] == pytest.approx(info.state.potential_energy, rel=1e-11)
            assert res["num_doublings"][0] == info.num_doublings
            assert res["n_leapfrog"][0] == info.n_leapfrog
            assert bool(res["is_turning"][0]) == info.is_turning
            assert res["acceptance_probability"][0] == pytest.approx(
                info.acceptance_probability, rel=1e-10)


def test_c_matches_numpy_hmc():
    D = 7
    r = np.random.default_rng(5)
    mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    nt, ct_ = no.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    m = co.Metric(imm, D)
    q0 = r.normal(size=D)
    kernel = no.hmc_kernel(no.RandomStream(21), nt)
    rng = co.site_states([21], 2)
    q, U, g = co.new_state(ct_, q0[None])
    st = no.new_state(q0, nt)
    for _ in range(5):
        info = kernel(st, 0.2, imm, 12)
        res = co.hmc_step(ct_, m, rng, 0.2, 12, q, U, g)
        st = info.state._replace(momentum=None)
        np.testing.assert_allclose(q[0], info.state.position, rtol=1e-12, atol=1e-14)
        assert res["acceptance_probability"][0] == pytest.approx(info.acceptance_probability, rel=1e-10)


# ------------------------------------------------------------------ the dense branch against the pinned diagonal branch
def _triangular_pair(D, seed):
    """A diagonal problem (DiagGaussian(mu, sigma), inverse mass diag m) and its image under q' = A q
    with A LOWER TRIANGULAR (positive diagonal): dense MVN(A mu, A^-T diag(sigma^-2) A^-1) with the
    dense inverse mass matrix A diag(m) A^T, whose Cholesky factor is exactly A diag(sqrt m)."""
    r = np.random.default_rng(seed)
    mu, sigma, m = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    A = np.diag(0.7 + 0.6 * r.random(D)) + 0.3 * np.tril(r.normal(size=(D, D)), -1) / np.sqrt(D)
    Ainv = np.linalg.inv(A)
    P = Ainv.T @ np.diag(1 / sigma ** 2) @ Ainv
    imm = A @ np.diag(m) @ A.T
    return mu, sigma, m, A, 0.5 * (P + P.T), 0.5 * (imm + imm.T)


@pytest.mark.parametrize("sampler", ["nuts", "hmc"])
def test_dense_branch_equals_diagonal_branch_under_triangular_map(sampler):
    """No reference value exists for a dense-metric trajectory.  But the whole transition
    is equivariant under q' = A q for lower-triangular A: the momentum L'^-T z of the mapped problem is
    A^-T applied to the diagonal problem's sqrt(1/m) z (same draws z), leapfrogs, energies, U-turn
    products and acceptance probabilities coincide, so the chains satisfy q'_t = A q_t with IDENTICAL
    tree shapes and RNG consumption -- in exact arithmetic, and to rounding here.  This ties every
    dense-branch operation of both restatements (Cholesky / L^-T momentum, dense velocity, kinetic
    energy, is_turning, dense-MVN gradient) to the diagonal branch, which the reference's golden values
    pin (G1, G2, the unit tables)."""
    D, C = 7, 5
    mu, sigma, m, A, P, imm = _triangular_pair(D, 1)
    q0 = np.random.default_rng(2).normal(size=(C, D))
    seeds = [10 + c for c in range(C)]
    ns = 4 if sampler == "nuts" else 2
    td, tm = co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma), co.Target(co.T_DENSE_MVN, D, mu=A @ mu, prec=P)
    md, mm = co.Metric(m, D), co.Metric(imm, D)
    rd, rm = co.site_states(seeds, ns), co.site_states(seeds, ns)
    qd, Ud, gd = co.new_state(td, q0.copy())
    qm, Um, gm = co.new_state(tm, (q0 @ A.T).copy())
    const = Ud[0] - Um[0]  # sum(log sigma) + D log sqrt(2 pi): the dense-MVN potential carries no constant
    for t in range(4):
        if sampler == "nuts":
            a = co.nuts_step(td, md, rd, 0.3, qd, Ud, gd, max_exp=7)
            b = co.nuts_step(tm, mm, rm, 0.3, qm, Um, gm, max_exp=7)
            assert a["num_doublings"].tolist() == b["num_doublings"].tolist()
            assert a["is_turning"].tolist() == b["is_turning"].tolist()
        else:
            a = co.hmc_step(td, md, rd, 0.3, 9, qd, Ud, gd)
            b = co.hmc_step(tm, mm, rm, 0.3, 9, qm, Um, gm)
        assert a["n_leapfrog"].tolist() == b["n_leapfrog"].tolist() and a["is_diverging"].tolist() == b["is_diverging"].tolist()
        np.testing.assert_allclose(qm, qd @ A.T, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(b["momentum"], a["momentum"] @ np.linalg.inv(A), rtol=1e-10, atol=1e-11)  # p' = A^-T p
        np.testing.assert_allclose(Um, Ud - const, rtol=1e-11)
        np.testing.assert_allclose(gm, gd @ np.linalg.inv(A), rtol=1e-10, atol=1e-11)                          # g' = A^-T g
        np.testing.assert_allclose(b["acceptance_probability"], a["acceptance_probability"], rtol=1e-11)
        assert np.array_equal(rd, rm)  # same RNG consumption at every site
    if sampler == "nuts":  # ... and the numpy restatement's dense branch on chain 0
        kern = no.nuts_kernel(no.RandomStream(seeds[0]), no.DenseMVN(A @ mu, P), max_num_expansions=7)
        st = no.new_state(A @ q0[0], no.DenseMVN(A @ mu, P))
        rd0 = co.site_states(seeds[:1], 4)
        q1, U1, g1 = co.new_state(td, q0[:1].copy())
        for t in range(3):
            info = kern(st, 0.3, imm)
            st = info.state._replace(momentum=None)
            res = co.nuts_step(td, md, rd0, 0.3, q1, U1, g1, max_exp=7)
            assert info.n_leapfrog == res["n_leapfrog"][0] and info.num_doublings == res["num_doublings"][0]
            np.testing.assert_allclose(info.state.position, A @ q1[0], rtol=1e-10, atol=1e-12)
