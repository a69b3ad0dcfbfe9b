"""The dense-metric branch against a SECOND, independent derivation (tests/golden/make_dense_pin.py ->
tests/golden/dense_pin_v1.json; written from SURVEY.md Appendix A with the literals of metrics.py:52-59, no
oracle import): both restatements on the CPU, the HIP path on the GPU.  Not a reference-generated value (Aesara is
absent here) -- a second pin beside the unit tables of tests/test_metrics.py and the triangular-map invariance."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = json.load(open(os.path.join(HERE, "golden", "dense_pin_v1.json")))["cases"]
NUTS = [c for c in PIN if c["sampler"] == "nuts"]
HMC = [c for c in PIN if c["sampler"] == "hmc"]


def _check_nuts(case, pos, U, grad, mom, acc, nd, turn, div, nl, rtol):
    e = case["expect"]
    np.testing.assert_allclose(pos, e["position"], rtol=rtol, atol=1e-13)
    np.testing.assert_allclose(U, e["U"], rtol=rtol)
    np.testing.assert_allclose(grad, e["grad"], rtol=rtol, atol=1e-12)
    np.testing.assert_allclose(mom, e["momentum"], rtol=rtol, atol=1e-13)
    np.testing.assert_allclose(acc, e["acceptance_probability"], rtol=rtol)
    assert (int(nd), bool(turn), bool(div), int(nl)) == (e["num_doublings"], e["is_turning"], e["is_diverging"], e["n_leapfrog"])


def test_the_pin_covers_the_branches_it_is_for():
    tr = {c["name"]: c["expect"] for c in NUTS}
    assert tr["nuts-dense-c"]["n_leapfrog"] == 36 and len({t["direction"] for t in tr["nuts-dense-c"]["trace"]}) == 2
    assert tr["nuts-dense-a"]["trace"][-1]["length"] < 5 and not tr["nuts-dense-a"]["is_turning"]  # sub-trajectory U-turn
    assert tr["nuts-dense-b"]["is_turning"] and tr["nuts-dense-cut"]["num_doublings"] == 3         # whole-trajectory / cut


@pytest.mark.parametrize("case", NUTS, ids=lambda c: c["name"])
def test_numpy_restatement_matches_the_independent_derivation(case):
    from oracle import np_oracle as no
    tgt = no.DenseMVN(np.array(case["mu"]), np.array(case["prec"]))
    kernel = no.nuts_kernel(no.RandomStream(case["seed"]), tgt, max_num_expansions=case["max_exp"])
    info = kernel(no.new_state(np.array(case["q0"]), tgt), case["eps"], np.array(case["imm"]))
    s = info.state
    _check_nuts(case, s.position, s.potential_energy, s.potential_energy_grad, s.momentum, info.acceptance_probability,
                info.num_doublings, info.is_turning, info.is_diverging, info.n_leapfrog, 1e-12)


@pytest.mark.parametrize("case", NUTS, ids=lambda c: c["name"])
def test_c_restatement_matches_the_independent_derivation(case):
    from oracle import c_oracle as co
    otgt = co.Target(co.T_DENSE_MVN, 3, mu=np.array(case["mu"]), prec=np.array(case["prec"]))
    q, U, g = co.new_state(otgt, np.array(case["q0"]))
    rng = co.site_states([case["seed"]], 4)
    res = co.nuts_step(otgt, co.Metric(np.array(case["imm"]), 3), rng, case["eps"], q, U, g, max_exp=case["max_exp"])
    _check_nuts(case, q[0], U[0], g[0], res["momentum"][0], res["acceptance_probability"][0], res["num_doublings"][0],
                res["is_turning"][0], res["is_diverging"][0], res["n_leapfrog"][0], 1e-11)


@pytest.mark.parametrize("case", HMC, ids=lambda c: c["name"])
def test_restatements_match_the_independent_hmc_derivation(case):
    from oracle import c_oracle as co
    from oracle import np_oracle as no
    e = case["expect"]
    otgt = co.Target(co.T_DENSE_MVN, 3, mu=np.array(case["mu"]), prec=np.array(case["prec"]))
    q, U, g = co.new_state(otgt, np.array(case["q0"]))
    res = co.hmc_step(otgt, co.Metric(np.array(case["imm"]), 3), co.site_states([case["seed"]], 2), case["eps"], case["L"], q, U, g)
    np.testing.assert_allclose(q[0], e["position"], rtol=1e-11)
    np.testing.assert_allclose(res["acceptance_probability"][0], e["acceptance_probability"], rtol=1e-11)
    assert bool(res["accepted"][0]) == e["accepted"] and bool(res["is_diverging"][0]) == e["is_diverging"]
    tgt = no.DenseMVN(np.array(case["mu"]), np.array(case["prec"]))
    kernel = no.hmc_kernel(no.RandomStream(case["seed"]), tgt)
    info = kernel(no.new_state(np.array(case["q0"]), tgt), case["eps"], np.array(case["imm"]), case["L"])
    np.testing.assert_allclose(info.state.position, e["position"], rtol=1e-12)
    np.testing.assert_allclose(info.acceptance_probability, e["acceptance_probability"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("linear", [1, 0])
@pytest.mark.parametrize("case", PIN, ids=lambda c: c["name"])
def test_hip_matches_the_independent_derivation(case, linear):
    """The product's dense path (fp64 MFMA GEMMs, L^-T from its own blocked Cholesky, both dense modes) with no
    oracle at run time."""
    import torch
    from aehmc_amd import RandomStream, hmc, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    eng.set_option("dense_linear", linear)
    try:
        tgt = targets.DenseMVN(np.array(case["mu"]), np.array(case["prec"]))
        e = case["expect"]
        if case["sampler"] == "nuts":
            kernel = nuts.new_kernel(RandomStream(seed=case["seed"]), tgt, max_num_expansions=case["max_exp"])
            info, _ = kernel(nuts.new_state(torch.as_tensor(np.array(case["q0"]), device="cuda"), tgt), case["eps"],
                             np.array(case["imm"]))
            s = info.state
            _check_nuts(case, s.position.cpu().numpy(), s.potential_energy.item(), s.potential_energy_grad.cpu().numpy(),
                        s.momentum.cpu().numpy(), info.acceptance_probability.item(), info.num_doublings.item(),
                        info.is_turning.item(), info.is_diverging.item(), info.n_leapfrog.item(), 1e-9)
        else:
            kernel = hmc.new_kernel(RandomStream(seed=case["seed"]), tgt)
            info, _ = kernel(hmc.new_state(torch.as_tensor(np.array(case["q0"]), device="cuda"), tgt), case["eps"],
                             np.array(case["imm"]), case["L"])
            np.testing.assert_allclose(info.state.position.cpu().numpy(), e["position"], rtol=1e-9)
            np.testing.assert_allclose(info.acceptance_probability.item(), e["acceptance_probability"], rtol=1e-9)
            assert bool(info.is_diverging.item()) == e["is_diverging"]
    finally:
        eng.set_option("dense_linear", 1)


def _independent():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_dense_pin", os.path.join(HERE, "golden", "make_dense_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("block", range(4))
def test_restatements_match_the_independent_derivation_on_random_cases(block):
    """Beyond the four committed cases: the independent derivation itself (tests/golden/make_dense_pin.py, imported
    here) against the C and numpy restatements on 40 random dense problems -- D = 2 ... 6, random SPD precision and
    inverse mass matrix, step sizes from tiny (deep trees, cut at max_num_expansions) to large (early U-turns, low
    acceptance), both samplers: every discrete output identical, values to 1e-10."""
    from oracle import c_oracle as co
    from oracle import np_oracle as no
    ind = _independent()
    r = np.random.default_rng(900 + block)
    depth_seen = set()
    for k in range(10):
        D = int(r.integers(2, 7))
        A, B = r.normal(size=(D, D)), r.normal(size=(D, D))
        P = A @ A.T / D + 0.5 * np.eye(D)
        imm = B @ B.T / D + 0.5 * np.eye(D)
        P, imm = 0.5 * (P + P.T), 0.5 * (imm + imm.T)
        mu, q0 = r.normal(size=D), r.normal(size=D)
        eps = float(np.exp(r.uniform(np.log(0.02), np.log(0.9))))
        max_exp = int(r.choice([3, 6, 10]))
        seed = int(r.integers(0, 2 ** 31))
        e = ind.nuts_transition(seed, q0, mu, P, imm, eps, max_exp)
        otgt = co.Target(co.T_DENSE_MVN, D, mu=mu, prec=P)
        q, U, g = co.new_state(otgt, q0.copy())
        res = co.nuts_step(otgt, co.Metric(imm, D), co.site_states([seed], 4), eps, q, U, g, max_exp=max_exp)
        np.testing.assert_allclose(q[0], e["position"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(res["acceptance_probability"][0], e["acceptance_probability"], rtol=1e-10)
        got = (int(res["n_leapfrog"][0]), int(res["num_doublings"][0]), bool(res["is_turning"][0]), bool(res["is_diverging"][0]))
        assert got == (e["n_leapfrog"], e["num_doublings"], e["is_turning"], e["is_diverging"]), (block, k)
        tgt = no.DenseMVN(mu, P)
        info = no.nuts_kernel(no.RandomStream(seed), tgt, max_num_expansions=max_exp)(no.new_state(q0, tgt), eps, imm)
        np.testing.assert_allclose(info.state.position, e["position"], rtol=1e-10, atol=1e-12)
        assert info.n_leapfrog == e["n_leapfrog"]
        depth_seen.add(e["num_doublings"])
        h = ind.hmc_transition(seed, q0, mu, P, imm, eps, 7)
        q, U, g = co.new_state(otgt, q0.copy())
        hres = co.hmc_step(otgt, co.Metric(imm, D), co.site_states([seed], 2), eps, 7, q, U, g)
        np.testing.assert_allclose(q[0], h["position"], rtol=1e-10, atol=1e-12)
        assert bool(hres["accepted"][0]) == h["accepted"]
    assert len(depth_seen) >= 3


@pytest.mark.gpu
def test_hip_matches_the_independent_derivation_on_random_cases():
    """The product's dense path against the independent derivation itself on 16 random dense problems (D = 2 ... 6,
    single chains, trees from a few to dozens of leapfrogs) -- no oracle involved."""
    import torch
    from aehmc_amd import RandomStream, nuts, targets
    ind = _independent()
    r = np.random.default_rng(4242)
    deep = 0
    for k in range(16):
        D = int(r.integers(2, 7))
        A, B = r.normal(size=(D, D)), r.normal(size=(D, D))
        P = A @ A.T / D + 0.5 * np.eye(D)
        imm = B @ B.T / D + 0.5 * np.eye(D)
        P, imm = 0.5 * (P + P.T), 0.5 * (imm + imm.T)
        mu, q0 = r.normal(size=D), r.normal(size=D)
        eps = float(np.exp(r.uniform(np.log(0.03), np.log(0.7))))
        max_exp = int(r.choice([4, 10]))
        seed = int(r.integers(0, 2 ** 31))
        e = ind.nuts_transition(seed, q0, mu, P, imm, eps, max_exp)
        tgt = targets.DenseMVN(mu, P)
        kernel = nuts.new_kernel(RandomStream(seed=seed), tgt, max_num_expansions=max_exp)
        info, _ = kernel(nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt), eps, imm)
        np.testing.assert_allclose(info.state.position.cpu().numpy(), e["position"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(info.acceptance_probability.item(), e["acceptance_probability"], rtol=1e-9)
        got = (info.n_leapfrog.item(), info.num_doublings.item(), bool(info.is_turning.item()), bool(info.is_diverging.item()))
        assert got == (e["n_leapfrog"], e["num_doublings"], e["is_turning"], e["is_diverging"]), k
        deep += e["n_leapfrog"] >= 10
    assert deep >= 4
