"""The adaptation algorithms as public building blocks -- ``aehmc_amd.algorithms.{dual_averaging, welford_covariance}``,
``mass_matrix.covariance_adaptation``, ``window_adaptation.window_adaptation`` -- with the reference's
``(init, update[, final])`` call shapes.  The first tests restate the reference's own
(/root/reference/tests/test_algorithms.py:11-133, tests/test_mass_matrix.py:11-60: same inputs, same expected values);
the rest hold the HIP path to the numpy restatement (oracle/np_adaptation.py) and to ``window_adaptation.run``."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import np_adaptation as na  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


def test_dual_averaging_minimises_a_quadratic():
    """tests/test_algorithms.py:10-57: gradient of (x - 1)^2, shrinkage point 0.5, 100 updates -> the iterate and
    its average are at 1."""
    from aehmc_amd import algorithms
    init, update = algorithms.dual_averaging(gamma=0.5)
    state = init(0.5)
    ref_init, ref_update = na.dual_averaging(gamma=0.5)
    ref = ref_init(0.5)
    for _ in range(100):
        x = state.iterates
        state = update(2 * (x - 1), state)
        ref = ref_update(2 * (ref.iterates - 1), ref)
    assert state.step.item() == 101 == ref.step
    assert state.iterates_avg.item() == pytest.approx(1.0, 1e-2) and state.iterates.item() == pytest.approx(1.0, 1e-2)
    assert state.iterates_avg.item() == pytest.approx(ref.iterates_avg, rel=1e-12)
    assert state.iterates.item() == pytest.approx(ref.iterates, rel=1e-12)


@pytest.mark.parametrize("num_dims", [0, 1, 3])
@pytest.mark.parametrize("do_compute_covariance", [True, False])
def test_welford_constant(num_dims, do_compute_covariance):
    """tests/test_algorithms.py:60-96"""
    from aehmc_amd import algorithms
    sample = dev(np.ones(num_dims)) if num_dims > 0 else dev(1.0)
    init, update, final = algorithms.welford_covariance(do_compute_covariance)
    state = init(num_dims)
    for _ in range(10):
        state = update(sample, *state)
    mean = state[0].cpu().numpy()
    if num_dims > 0:
        assert mean.shape == (num_dims,)
        np.testing.assert_allclose(mean, np.ones(num_dims), rtol=1e-1)
    else:
        assert mean.ndim == 0 and mean == 1.0
    cov = final(state[1], state[2]).cpu().numpy()
    if num_dims > 0:
        expected = np.zeros((num_dims, num_dims)) if do_compute_covariance else np.zeros(num_dims)
        assert cov.shape == expected.shape
        np.testing.assert_allclose(cov, expected)
    else:
        assert cov.ndim == 0 and cov == 0


@pytest.mark.parametrize("do_compute_covariance", [True, False])
@pytest.mark.parametrize("n_dim", [1, 3])
def test_welford(n_dim, do_compute_covariance):
    """tests/test_algorithms.py:99-119: samples 0 .. 9 -> mean 9/2, variance 55/6"""
    from aehmc_amd import algorithms
    init, update, final = algorithms.welford_covariance(do_compute_covariance)
    state = init(n_dim)
    for i in range(10):
        state = update(dev(i * np.ones(n_dim)), *state)
    np.testing.assert_allclose(state[0].cpu().numpy(), (9.0 / 2) * np.ones(n_dim))
    cov = final(state[1], state[2]).cpu().numpy()
    expected = 55.0 / 6.0 * (np.ones((n_dim, n_dim)) if do_compute_covariance else np.ones(n_dim))
    assert cov.shape == expected.shape
    np.testing.assert_allclose(cov, expected)


@pytest.mark.parametrize("do_compute_covariance", [True, False])
def test_welford_scalar(do_compute_covariance):
    """tests/test_algorithms.py:122-133"""
    from aehmc_amd import algorithms
    init, update, final = algorithms.welford_covariance(do_compute_covariance)
    state = init(0)
    for i in range(10):
        state = update(dev(float(i)), *state)
    cov = final(state[1], state[2]).cpu().numpy()
    assert cov.ndim == 0 and cov == pytest.approx(55.0 / 6.0)


@pytest.mark.parametrize("is_full_matrix", [True, False])
@pytest.mark.parametrize("n_dims", [0, 1, 3])
def test_mass_matrix_adaptation(is_full_matrix, n_dims):
    """tests/test_mass_matrix.py:11-60: 2000 draws from N(0.5, 0.33 * ones) -> the adapted inverse mass matrix is the
    covariance (its diagonal) within 10 %.  (Draws from numpy here: the reference draws them inside its scan.)"""
    from aehmc_amd import mass_matrix
    r = np.random.default_rng(0)
    if n_dims > 0:
        cov = 0.33 * np.ones((n_dims, n_dims))
        draws = r.multivariate_normal(0.5 * np.ones(n_dims), cov, size=2000, method="svd")
    else:
        cov = 0.33
        draws = r.normal(0.5, cov, size=2000)
    init, update, final = mass_matrix.covariance_adaptation(is_full_matrix)
    imm0, wc_state = init(n_dims)
    assert tuple(imm0.shape) == (() if n_dims == 0 else ((n_dims, n_dims) if is_full_matrix else (n_dims,)))
    ref_init, ref_update, ref_final = na.covariance_adaptation(is_full_matrix)
    _, ref_state = ref_init(n_dims)
    for x in draws:
        wc_state = update(dev(x), wc_state)
        ref_state = ref_update(np.asarray(x), ref_state)
    imm = final(wc_state).cpu().numpy()
    np.testing.assert_allclose(imm, ref_final(ref_state), rtol=1e-10)   # the numpy restatement, tightly
    if n_dims > 0:
        expected = cov if is_full_matrix else np.diagonal(cov)
        assert imm.shape == np.shape(expected)
        np.testing.assert_allclose(imm, expected, rtol=0.1)
    else:
        assert np.ndim(imm) == 0 and np.sqrt(imm) == pytest.approx(cov, rel=0.1)


def test_building_blocks_for_many_chains_at_once():
    """``num_chains=C`` (not in the reference, which has no chain axis): C independent estimators in one launch equal
    C single estimators."""
    from aehmc_amd import algorithms, mass_matrix
    r = np.random.default_rng(3)
    C, D, n = 5, 70, 40
    xs = r.normal(size=(n, C, D)) * (1 + np.arange(D))
    for full in (False, True):
        init, update, final = mass_matrix.covariance_adaptation(full, num_chains=C)
        _, st = init(D)
        singles = []
        one_init, one_update, one_final = mass_matrix.covariance_adaptation(full)
        for c in range(C):
            singles.append(one_init(D)[1])
        for t in range(n):
            st = update(dev(xs[t]), st)
            singles = [one_update(dev(xs[t, c]), s) for c, s in enumerate(singles)]
        out = final(st)
        assert tuple(out.shape) == ((C, D, D) if full else (C, D)) and st[2].tolist() == [n] * C
        for c in range(C):
            assert torch.equal(out[c], one_final(singles[c]))
    winit, wupdate, wfinal = algorithms.welford_covariance(True, num_chains=C)
    s = winit(0)  # scalar problems, one per chain
    for t in range(n):
        s = wupdate(dev(xs[t, :, 0]), *s)
    np.testing.assert_allclose(wfinal(s[1], s[2]).cpu().numpy(), xs[:, :, 0].var(axis=0, ddof=1), rtol=1e-12)


@pytest.mark.parametrize("full", [False, True])
def test_window_adaptation_init_update_equals_run(full):
    """Driving the warm-up with ``(init, update)`` around the kernel (the reference's window_adaptation.py:119-227
    shape) gives bit for bit what ``window_adaptation.run`` gives -- the fused one-call warm-up and its step-by-step
    loop alike -- and leaves earlier states untouched (states are values)."""
    from aehmc_amd import RandomStream, nuts, targets, window_adaptation
    r = np.random.default_rng(8)
    C, D, n = 6, 5, 120
    mu, sigma = r.normal(size=D), 0.3 + 2 * r.random(D)
    tgt = targets.DiagGaussian(mu, sigma)
    q0 = r.normal(size=(C, D))
    seeds = list(range(900, 900 + C))

    kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=6)
    state = nuts.new_state(dev(q0), tgt)
    init, update = window_adaptation.window_adaptation(n, is_mass_matrix_full=full, initial_step_size=0.5)
    ws, params = init(state)
    first = ws
    first_eps = ws.step_size.clone()
    for i in range(n):
        info, _ = kern(state, *params)
        state = info.state._replace(momentum=None)
        ws, params = update(i, ws, params, info)
    assert torch.equal(first.step_size, first_eps) and first.da_state.step.tolist() == [1] * C

    for fused in (True, False):
        kern2 = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=6)
        state2, (eps2, imm2), _ = window_adaptation.run(kern2, nuts.new_state(dev(q0), tgt), n, is_mass_matrix_full=full,
                                                        initial_step_size=0.5, fused=fused)
        assert torch.equal(state2.position, state.position)
        assert torch.equal(eps2.value, params[0].value) and torch.equal(imm2.value, params[1].value)
    assert tuple(params[1].value.shape) == ((C, D, D) if full else (C, D))
