"""NUTS with one DENSE inverse mass matrix PER CHAIN at 64 < D <= 512 in one launch (csrc/nuts_pc_dense.cuh): what
``window_adaptation.run(is_mass_matrix_full=True)`` returns (/root/reference/aehmc/mass_matrix.py:12-120,
window_adaptation.py:119-227: adaptation is per chain).  Chain c against the C restatement run with matrix c (1e-9,
discrete outputs identical) and BITWISE against the lock-step path (option pc_dense = 0), single transitions and
kernel.sample(T)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle as co  # noqa: E402


def problem(D, C, seed):
    r = np.random.default_rng(seed)
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    A = r.normal(size=(C, D, D))
    imm = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    imm = 0.5 * (imm + imm.transpose(0, 2, 1))
    eps = 0.3 * (0.5 + r.random(C)) * (5 / D) ** 0.25
    return mu, sigma, imm, eps, r.normal(size=(C, D))


@pytest.fixture()
def eng():
    from aehmc_amd.engine import get_engine
    e = get_engine()
    try:
        yield e
    finally:
        e.set_option("pc_dense", 1)


@pytest.mark.parametrize("D", [65, 128, 200, 300, 512])
def test_per_chain_dense_nuts_matches_oracle_and_lockstep_bitwise(eng, D):
    from aehmc_amd import PerChain, RandomStream, nuts, targets
    C = 6
    mu, sigma, imm, eps, q0 = problem(D, C, D)
    seeds = [70 + c for c in range(C)]
    tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    runs = []
    for pc in (1, 0):
        eng.set_option("pc_dense", pc)
        kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=7)
        state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        outs = []
        for _ in range(2):
            info, _ = kernel(state, PerChain(torch.as_tensor(eps, device="cuda")), PerChain(torch.as_tensor(imm, device="cuda")))
            state = info.state._replace(momentum=None)
            outs.append(info)
        runs.append(outs)
    for a, b in zip(*runs):  # one launch per call == the lock-step path, bit for bit
        assert torch.equal(a.state.position, b.state.position) and torch.equal(a.n_leapfrog, b.n_leapfrog)
        assert torch.equal(a.acceptance_probability, b.acceptance_probability)
        assert torch.equal(a.state.potential_energy_grad, b.state.potential_energy_grad)
    for c in range(C):
        rng = co.site_states([seeds[c]], 4)
        metric = co.Metric(imm[c], D)
        q, U, g = co.new_state(otgt, q0[c:c + 1].copy())
        for info in runs[0]:
            res = co.nuts_step(otgt, metric, rng, float(eps[c]), q, U, g, max_exp=7)
            assert info.n_leapfrog[c].item() == res["n_leapfrog"][0]
            np.testing.assert_allclose(info.state.position[c].cpu().numpy(), q[0], rtol=1e-9, atol=1e-11)
            assert info.acceptance_probability[c].item() == pytest.approx(res["acceptance_probability"][0], rel=1e-9)


def test_per_chain_dense_sample_in_one_launch_equals_repeated_steps(eng):
    from aehmc_amd import PerChain, RandomStream, nuts, targets
    D, C, T = 150, 9, 4
    mu, sigma, imm, eps, q0 = problem(D, C, 7)
    tgt = targets.DiagGaussian(mu, sigma)
    pe, pi = PerChain(torch.as_tensor(eps, device="cuda")), PerChain(torch.as_tensor(imm, device="cuda"))
    k1 = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    samples, info = k1.sample(nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt), pe, pi, T)[:2]
    k2 = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    state, total = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt), 0
    for t in range(T):
        one, _ = k2(state, pe, pi)
        state = one.state._replace(momentum=None)
        total = total + one.n_leapfrog
        assert torch.equal(samples[t], one.state.position)
    assert torch.equal(info.n_leapfrog, total) and torch.equal(info.state.position, state.position)


@pytest.mark.parametrize("D", [70, 200])
def test_per_chain_dense_hmc_matches_oracle_and_lockstep_bitwise(eng, D):
    from aehmc_amd import PerChain, RandomStream, hmc, targets
    C, L = 5, 6
    mu, sigma, imm, eps, q0 = problem(D, C, D + 1)
    seeds = [30 + c for c in range(C)]
    tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    pe = PerChain(torch.as_tensor(eps, device="cuda"))
    pi = PerChain(torch.as_tensor(imm, device="cuda"))
    runs = []
    for pc in (1, 0):
        eng.set_option("pc_dense", pc)
        kernel = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
        samples, info = kernel.sample(hmc.new_state(torch.as_tensor(q0, device="cuda"), tgt), pe, pi, L, 3)[:2]
        runs.append((samples, info))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1].acceptance_probability, runs[1][1].acceptance_probability)
    for c in range(C):
        rng = co.site_states([seeds[c]], 2)
        metric = co.Metric(imm[c], D)
        q, U, g = co.new_state(otgt, q0[c:c + 1].copy())
        for t in range(3):
            res = co.hmc_step(otgt, metric, rng, float(eps[c]), L, q, U, g)
            np.testing.assert_allclose(runs[0][0][t, c].cpu().numpy(), q[0], rtol=1e-9, atol=1e-11)
        assert runs[0][1].acceptance_probability[c].item() == pytest.approx(res["acceptance_probability"][0], rel=1e-9)


def test_per_chain_factors_are_kept_for_an_unchanged_device_tensor(eng):
    """A bare [C, D, D] PerChain metric is factored (L^-T per chain, one wavefront per matrix) when it is first bound,
    not at every call: the factors are kept while the device tensor is unchanged (identity + version counter)."""
    from aehmc_amd import PerChain, RandomStream, nuts, targets
    D, C = 80, 8
    mu, sigma, imm, eps, q0 = problem(D, C, 3)
    t_imm = torch.as_tensor(imm, device="cuda")
    tgt = targets.DiagGaussian(mu, sigma)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=5)
    state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    n0 = eng.n_metric_factorizations
    for _ in range(3):
        info, _ = kernel(state, 0.1, PerChain(t_imm))
        state = info.state._replace(momentum=None)
    assert eng.n_metric_factorizations == n0 + 1
    t_imm.mul_(1.5)  # an in-place edit is seen
    kernel(state, 0.1, PerChain(t_imm))
    assert eng.n_metric_factorizations == n0 + 2
