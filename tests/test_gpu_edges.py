"""Edge cases through the C-ABI: ragged sizes, degenerate arguments, non-finite inputs,
error paths (reference behaviours: metrics.py:60-63, proposals.py:43-45, hmc.py:189-191)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle as co  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda")


@pytest.mark.parametrize("C,D", [(1, 1), (3, 63), (5, 64), (7, 65), (2, 129), (9, 1025)])
def test_ragged_shapes_nuts_and_hmc(C, D):
    """C not a multiple of 4 chains per block, D around the 64-lane and register-tile edges
    (1025 > the fused HMC kernel's limit -> lock-step path)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(C * 1000 + D)
    mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    metric = co.Metric(imm, D)
    seeds = list(range(C))
    q0 = r.normal(size=(C, D))
    eps = 0.3 / D ** 0.25
    srng = RandomStream(seeds=seeds)
    nk, hk = nuts.new_kernel(srng, tgt, max_num_expansions=5), hmc.new_kernel(srng, tgt)
    state = nuts.new_state(dev(q0), tgt)
    q, U, g = co.new_state(otgt, q0.copy())
    rng, hrng = co.site_states(seeds, 4), co.site_states(seeds, 2, first_site=4)
    info, _ = nk(state, eps, imm)
    co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=5)
    np.testing.assert_allclose(info.state.position.cpu().numpy(), q, rtol=1e-9, atol=1e-12)
    info, _ = hk(info.state._replace(momentum=None), eps, imm, 5)
    co.hmc_step(otgt, metric, hrng, eps, 5, q, U, g)
    np.testing.assert_allclose(info.state.position.cpu().numpy(), q, rtol=1e-9, atol=1e-12)


def test_degenerate_arguments():
    from aehmc_amd import RandomStream, hmc, nuts, targets
    tgt = targets.StdNormal()
    q0 = dev(np.random.default_rng(0).normal(size=(4, 3)))
    state = hmc.new_state(q0, tgt)
    # L = 0: nothing integrates, delta = 0 -> p_accept = 1, state unchanged (hmc.py:185-195)
    info, _ = hmc.new_kernel(RandomStream(seeds=[1, 2, 3, 4]), tgt)(state, 0.1, np.ones(3), 0)
    assert torch.equal(info.state.position, q0) and (info.acceptance_probability == 1).all()
    assert (info.n_leapfrog == 0).all()
    # a single expansion: 2 leapfrogs (2**0 + 1)
    info, _ = nuts.new_kernel(RandomStream(seeds=[1, 2, 3, 4]), tgt, max_num_expansions=1)(state, 1e-3, np.ones(3))
    assert (info.n_leapfrog == 2).all() and (info.num_doublings == 1).all()
    # negative step size integrates backwards in time: still a valid transition
    info, _ = nuts.new_kernel(RandomStream(seeds=[1, 2, 3, 4]), tgt)(state, -0.2, np.ones(3))
    assert torch.isfinite(info.state.position).all()


def test_non_finite_energy_is_divergence_not_error():
    """NaN energy -> delta = -inf -> is_diverging, p_accept = 0, chain stays (hmc.py:189-195;
    proposals.py:43-45)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    tgt = targets.StdNormal()
    q0 = np.array([[0.5, 0.5], [1.0, -1.0]])
    state = hmc.new_state(dev(q0), tgt)
    info, _ = hmc.new_kernel(RandomStream(seeds=[0, 1]), tgt)(state, 1e200, np.ones(2), 3)
    assert info.is_diverging.all() and (info.acceptance_probability == 0).all()
    assert torch.equal(info.state.position.cpu(), torch.as_tensor(q0))
    info, _ = nuts.new_kernel(RandomStream(seeds=[0, 1]), tgt)(state, 1e200, np.ones(2))
    assert info.is_diverging.all() and torch.equal(info.state.position.cpu(), torch.as_tensor(q0))
    assert (info.num_doublings == 1).all()


def test_error_paths():
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import EngineError, get_engine
    tgt = targets.StdNormal()
    state = nuts.new_state(dev(np.zeros((2, 3))), tgt)
    kernel = nuts.new_kernel(RandomStream(seeds=[0, 1]), tgt)
    with pytest.raises(ValueError):  # metrics.py:60-63
        kernel(state, 0.1, np.ones((3, 3, 3)))
    with pytest.raises(ValueError):  # wrong diagonal length
        kernel(state, 0.1, np.ones(4))
    with pytest.raises(ValueError):  # non-symmetric dense imm
        kernel(state, 0.1, np.array([[1.0, 0.5, 0], [0.0, 1, 0], [0, 0, 1.0]]))
    with pytest.raises(ValueError):  # chain count mismatch
        kernel(nuts.new_state(dev(np.zeros((3, 3))), tgt), 0.1, np.ones(3))
    eng = get_engine()
    with pytest.raises(EngineError):
        eng.set_option("no_such_option", 1)
    bad = nuts.new_kernel(RandomStream(seeds=[0, 1]), tgt, max_num_expansions=0)
    with pytest.raises(EngineError):
        bad(state, 0.1, np.ones(3))


# ------------------------------------------------------------------ per-chain parameters
@pytest.mark.parametrize("kind,D", [("hmc", 70), ("hmc", 1500), ("nuts", 70), ("nuts", 600), ("nuts-linreg", 2)])
def test_per_chain_parameters_equal_single_chain_runs(kind, D):
    """PerChain step sizes and diagonal inverse mass matrices (what per-chain window adaptation
    produces; the reference runs one chain per compiled function): chain c of a batch equals
    chain c run alone with its own scalar step size and vector -- on the fused / wide HMC
    kernels, the lock-step and resident NUTS kernels and the regression workgroups."""
    from aehmc_amd import PerChain, RandomStream, hmc, nuts, targets
    r = np.random.default_rng(D)
    C = 5
    if kind == "nuts-linreg":
        X = r.normal(size=4000)
        tgt = targets.LinearRegression(X, 3 * X + r.normal(size=4000))
        q0 = np.array([3.0, 0.0]) + 0.01 * r.normal(size=(C, 2))
        eps = 0.02 * (0.5 + r.random(C))
        imm = 1e-3 * (0.5 + r.random((C, 2)))
    else:
        tgt = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
        q0 = r.normal(size=(C, D))
        eps = 0.3 / D ** 0.25 * (0.5 + r.random(C))
        imm = 0.5 + r.random((C, D))
    seeds = [900 + c for c in range(C)]
    mod = hmc if kind == "hmc" else nuts
    extra = (9,) if kind == "hmc" else ()

    def run(seed_list, q, e, m):
        kernel = mod.new_kernel(RandomStream(seeds=seed_list), tgt)
        state = mod.new_state(torch.as_tensor(q, device="cuda"), tgt)
        outs = []
        for _ in range(3):
            info, _ = kernel(state, e, m, *extra)
            state = info.state._replace(momentum=None)
            outs.append((info.state.position.clone(), info.acceptance_probability.clone()))
        return outs

    batch = run(seeds, q0, PerChain(torch.as_tensor(eps, device="cuda")), PerChain(torch.as_tensor(imm, device="cuda")))
    for c in range(C):
        single = run([seeds[c]], q0[c:c + 1], float(eps[c]), torch.as_tensor(imm[c], device="cuda"))
        for (qb, ab), (qs, a_s) in zip(batch, single):
            assert torch.equal(qb[c], qs[0]) and torch.equal(ab[c], a_s[0])


@pytest.mark.parametrize("D", [256, 257, 512, 513, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 8193, 10176, 10240, 10241])
def test_kernel_family_boundaries_at_large_d(D):
    """D at the edges of the wide-HMC (1025..10240) and resident-NUTS (..10176, LDS-bound) ranges:
    each side of a boundary runs (resident or lock-step) and agrees with the other path."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    C = 3
    tgt = targets.IsoGaussian()
    imm = torch.ones(D, dtype=torch.float64, device="cuda")
    q0 = torch.as_tensor(np.random.default_rng(D).standard_normal((C, D)), device="cuda")
    try:
        for mod, extra in ((hmc, (5,)), (nuts, ())):
            outs = []
            for opt in (1, 0):
                eng.set_option("fused_hmc", opt)
                eng.set_option("resident_nuts", 2 if opt else 0)
                k = mod.new_kernel(RandomStream(seeds=[1, 2, 3]), tgt)
                info, _ = k(mod.new_state(q0, tgt), 0.05, imm, *extra)
                outs.append(info)
            np.testing.assert_allclose(outs[0].state.position.cpu().numpy(), outs[1].state.position.cpu().numpy(),
                                       rtol=1e-12, atol=1e-14)
            assert torch.equal(outs[0].n_leapfrog, outs[1].n_leapfrog)
    finally:
        eng.set_option("fused_hmc", 1)
        eng.set_option("resident_nuts", 2)


@pytest.mark.parametrize("C,D", [(7, 3), (8, 3), (2048, 3), (2049, 3), (16384, 2), (16383, 100)])
def test_auto_kernel_choice_boundaries_in_chain_count(C, D):
    """The automatic resident / lock-step choice changes at 8, 2048 and 16384 chains: both sides of
    each boundary give what the lock-step path gives."""
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    tgt = targets.DiagGaussian(np.linspace(-1, 1, D), np.linspace(0.5, 2.0, D))
    imm = torch.ones(D, dtype=torch.float64, device="cuda")
    q0 = torch.as_tensor(np.random.default_rng(C).standard_normal((C, D)), device="cuda")
    outs = []
    try:
        for opt in (2, 0):
            eng.set_option("resident_nuts", opt)
            k = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            info, _ = k(nuts.new_state(q0, tgt), 0.3, imm)
            outs.append(info)
    finally:
        eng.set_option("resident_nuts", 2)
    np.testing.assert_allclose(outs[0].state.position.cpu().numpy(), outs[1].state.position.cpu().numpy(),
                               rtol=1e-12, atol=1e-14)
    assert torch.equal(outs[0].n_leapfrog, outs[1].n_leapfrog)
    assert torch.equal(outs[0].is_turning, outs[1].is_turning)


def test_per_chain_parameters_must_match_chain_count():
    """Per-chain step sizes / mass matrices of a different chain count (e.g. adaptation output
    reused with another C) are refused instead of being indexed out of bounds, and a per-chain
    step size does not leak into the next call of the shared engine."""
    from aehmc_amd import PerChain, RandomStream, hmc, nuts, targets
    from aehmc_amd.engine import EngineError
    tgt = targets.StdNormal()
    C, D = 6, 5
    q0 = dev(np.random.default_rng(0).normal(size=(C, D)))
    state = nuts.new_state(q0, tgt)
    nk = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    hk = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    with pytest.raises(EngineError, match="per-chain step sizes"):
        nk(state, PerChain(np.full(C + 2, 0.1)), np.ones(D))
    with pytest.raises(EngineError, match="per-chain step sizes"):
        hk(state, PerChain(np.full(C - 1, 0.1)), np.ones(D), 3)
    with pytest.raises(EngineError, match="per-chain inverse mass matrix"):
        nk(state, 0.1, PerChain(np.ones((C + 1, D))))
    # after a failed / per-chain call, a scalar-step call of another chain count works
    info, _ = nk(state, PerChain(np.full(C, 0.1)), PerChain(np.ones((C, D))))
    assert torch.isfinite(info.state.position).all()
    st2 = nuts.new_state(dev(np.zeros((2, D))), tgt)
    info, _ = nuts.new_kernel(RandomStream(seeds=[0, 1]), tgt)(st2, 0.1, np.ones(D))
    assert torch.isfinite(info.state.position).all()


def test_numpy_mass_matrix_edited_in_place_is_seen():
    """A large (> 65536 elements) numpy inverse mass matrix edited in place between calls must be
    re-read (the reference reads the matrix on every call); near-symmetric estimates (A @ A.T) are
    accepted as the reference's cholesky (one triangle) would."""
    from aehmc_amd import RandomStream, hmc, targets
    D, C = 300, 3
    r = np.random.default_rng(5)
    A = r.normal(size=(D, D)) / np.sqrt(D)
    imm = A @ A.T + np.eye(D)          # symmetric up to rounding only
    assert D * D > 65536
    tgt = targets.StdNormal()
    q0 = r.normal(size=(C, D))
    outs = []
    for scale in (1.0, 4.0):
        imm_use = imm if scale == 1.0 else imm.__imul__(scale)  # same object, edited in place
        assert imm_use is imm
        state = hmc.new_state(dev(q0), tgt)
        info, _ = hmc.new_kernel(RandomStream(seeds=[1, 2, 3]), tgt)(state, 0.05, imm, 4)
        outs.append(info.state.position.cpu().numpy())
    assert not np.allclose(outs[0], outs[1])
    # the second result equals a fresh engine-independent evaluation with the scaled matrix
    otgt, metric = co.Target(co.T_STD_NORMAL, D), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    co.hmc_step(otgt, metric, co.site_states([1, 2, 3], 2), 0.05, 4, q, U, g)
    np.testing.assert_allclose(outs[1], q, rtol=1e-9, atol=1e-12)
