"""Mid-size dense problems (shared dense inverse mass matrix, 64 < D <= 512) on the block-resident kernels
(csrc/nuts_block.cuh): one workgroup per 16 chains runs the whole call in one launch.  Two bars:
  * parity with the CPU oracle on identical seeds (RTOL 1e-9, every discrete output and the RNG state identical);
  * BITWISE equality with the lock-step path (option "block_dense" = 0), whose chain-batched fp64 GEMM the in-block
    MFMA products reproduce k-step for k-step -- whichever chains share a workgroup.
Reference: /root/reference/aehmc/metrics.py:52-73,94-102 (dense products), nuts.py:56-153, hmc.py:77-204."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle as co  # noqa: E402

RTOL = 1e-9


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


@pytest.fixture()
def eng():
    from aehmc_amd.engine import get_engine
    e = get_engine()
    e.set_option("block_dense", 1)
    try:
        yield e
    finally:
        e.set_option("block_dense", 1)


def make(tkind, D, r):
    from aehmc_amd import targets
    mu = r.normal(size=D)
    if tkind == "dense":
        A = r.normal(size=(D, D))
        prec = np.linalg.inv(A @ A.T / D + np.eye(D))
        prec = 0.5 * (prec + prec.T)
        tgt, otgt = targets.DenseMVN(mu, prec), co.Target(co.T_DENSE_MVN, D, mu=mu, prec=prec)
    elif tkind == "diag":
        sigma = 0.5 + r.random(D)
        tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    else:
        tgt, otgt = targets.StdNormal(), co.Target(co.T_STD_NORMAL, D)
    B = r.normal(size=(D, D))
    imm = B @ B.T / D + np.eye(D)
    return tgt, otgt, 0.5 * (imm + imm.T)


def check(info, q, U, g, res, nuts):
    np.testing.assert_allclose(info.state.position.cpu().numpy(), q, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.state.potential_energy.cpu().numpy(), U, rtol=RTOL)
    np.testing.assert_allclose(info.state.potential_energy_grad.cpu().numpy(), g, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.state.momentum.cpu().numpy(), res["momentum"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.acceptance_probability.cpu().numpy(), res["acceptance_probability"], rtol=RTOL)
    assert np.array_equal(info.is_diverging.cpu().numpy(), res["is_diverging"])
    assert np.array_equal(info.n_leapfrog.cpu().numpy(), res["n_leapfrog"])
    if nuts:
        assert np.array_equal(info.num_doublings.cpu().numpy(), res["num_doublings"])
        assert np.array_equal(info.is_turning.cpu().numpy(), res["is_turning"])


def same_bits(a, b):
    for name in ("position", "potential_energy", "potential_energy_grad", "momentum"):
        assert torch.equal(getattr(a.state, name), getattr(b.state, name)), name
    for name in ("acceptance_probability", "is_diverging", "n_leapfrog"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


CASES = [("dense", 65, 20), ("dense", 100, 33), ("diag", 130, 7), ("std", 200, 16), ("dense", 256, 19),
         ("dense", 333, 18), ("dense", 512, 17), ("diag", 512, 5)]


@pytest.mark.parametrize("tkind,D,C", CASES)
def test_block_dense_nuts_matches_oracle_and_lockstep_bitwise(eng, tkind, D, C):
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(D * 3 + C)
    tgt, otgt, imm = make(tkind, D, r)
    eps, max_exp = 0.9 * D ** -0.25, 6
    seeds = [2000 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    immd = dev(imm)

    def run(block):
        eng.set_option("block_dense", block)
        srng = RandomStream(seeds=seeds)
        kern = nuts.new_kernel(srng, tgt, max_num_expansions=max_exp)
        state = nuts.new_state(dev(q0), tgt)
        infos = []
        for _ in range(3):
            info, upd = kern(state, eps, immd)
            infos.append((info, upd[srng].clone()))
            state = info.state._replace(momentum=None)
        return infos

    # 1: chain state in registers up to D = 256 (work rows above); 2: work rows at every D; 0: the lock-step path
    blk, rows, lock = run(1), run(2), run(0)
    rng, metric = co.site_states(seeds, 4), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    for (ib, rb), (iw, rw), (il, rl) in zip(blk, rows, lock):
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp)
        check(ib, q, U, g, res, True)
        assert np.array_equal(rb.cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        for other, ro in ((iw, rw), (il, rl)):
            same_bits(ib, other)
            assert torch.equal(ib.num_doublings, other.num_doublings) and torch.equal(ib.is_turning, other.is_turning)
            assert torch.equal(rb, ro)
    assert int(sum(i.n_leapfrog.sum() for i, _ in blk)) > 3 * C * 3  # trees of more than one expansion


@pytest.mark.parametrize("tkind,D,C,L", [("dense", 100, 19, 9), ("diag", 257, 16, 5), ("dense", 400, 3, 4), ("std", 65, 40, 12)])
def test_block_dense_hmc_matches_oracle_and_lockstep_bitwise(eng, tkind, D, C, L):
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D * 5 + C)
    tgt, otgt, imm = make(tkind, D, r)
    eps = 0.6 * D ** -0.25
    seeds = [3000 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    immd = dev(imm)

    def run(block):
        eng.set_option("block_dense", block)
        srng = RandomStream(seeds=seeds)
        kern = hmc.new_kernel(srng, tgt)
        state = hmc.new_state(dev(q0), tgt)
        infos = []
        for _ in range(3):
            info, upd = kern(state, eps, immd, L)
            infos.append((info, upd[srng].clone()))
            state = info.state._replace(momentum=None)
        return infos

    blk, rows, lock = run(1), run(2), run(0)
    rng, metric = co.site_states(seeds, 2), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    for (ib, rb), (iw, rw), (il, rl) in zip(blk, rows, lock):
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        check(ib, q, U, g, res, False)
        assert np.array_equal(rb.cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        same_bits(ib, iw)
        same_bits(ib, il)
        assert torch.equal(rb, rl) and torch.equal(rb, rw)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("sampler", ["nuts", "hmc"])
def test_block_dense_sample_equals_repeated_steps(eng, sampler, mode):
    """kernel.sample(T) runs the T transitions of a workgroup's chains in one launch: per-transition positions,
    acceptance, divergence, leapfrog totals, final state and RNG state equal T separate calls bit for bit."""
    from aehmc_amd import RandomStream, hmc, nuts
    eng.set_option("block_dense", mode)
    r = np.random.default_rng(9)
    D, C, T = 96, 21, 4
    tgt, _, imm = make("dense", D, r)
    immd, q0 = dev(imm), r.normal(size=(C, D))
    seeds = list(range(70, 70 + C))
    mod = nuts if sampler == "nuts" else hmc
    extra = () if sampler == "nuts" else (7,)

    def kernel():
        srng = RandomStream(seeds=seeds)
        return srng, (mod.new_kernel(srng, tgt, max_num_expansions=5) if sampler == "nuts" else mod.new_kernel(srng, tgt))

    srng, kern = kernel()
    state = mod.new_state(dev(q0), tgt)
    out = kern.sample(state, 0.3, immd, *extra, T)
    samples, info, acc, div = out[0], out[1], out[2], out[3]
    srng2, kern2 = kernel()
    state = mod.new_state(dev(q0), tgt)
    total = torch.zeros(C, dtype=torch.int64, device="cuda")
    for t in range(T):
        i2, upd = kern2(state, 0.3, immd, *extra)
        assert torch.equal(samples[t], i2.state.position), t
        assert torch.equal(acc[t], i2.acceptance_probability) and torch.equal(div[t].bool(), i2.is_diverging.bool())
        total += i2.n_leapfrog
        state = i2.state._replace(momentum=None)
    assert torch.equal(info.state.position, i2.state.position)
    assert torch.equal(info.state.potential_energy, i2.state.potential_energy)
    assert torch.equal(info.n_leapfrog, total)
    assert torch.equal(kern._nuts["holder"]["rng"] if sampler == "nuts" else kern._hmc["holder"]["rng"],
                       kern2._nuts["holder"]["rng"] if sampler == "nuts" else kern2._hmc["holder"]["rng"])


@pytest.mark.parametrize("tkind,D,C,T", [("dense", 200, 37, 9), ("diag", 130, 16, 7), ("dense", 70, 5, 12), ("std", 256, 20, 6)])
def test_block_dense_chains_roll_on_without_changing_a_bit(eng, tkind, D, C, T):
    """sample(T) in the register kernel: a chain whose tree has ended begins its next transition while its neighbours
    are still in theirs ("block_roll": how many waiting chains trigger a begin round; 16 = transition by transition).
    Every schedule gives the bits of T separate calls on the lock-step path, RNG state included."""
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(D + C)
    tgt, _, imm = make(tkind, D, r)
    immd, q0 = dev(imm), r.normal(size=(C, D))
    seeds = list(range(900, 900 + C))
    eps = 0.5 * D ** -0.25

    def sample(roll):
        eng.set_option("block_dense", 1)
        eng.set_option("block_roll", roll)
        try:
            srng = RandomStream(seeds=seeds)
            kern = nuts.new_kernel(srng, tgt, max_num_expansions=6)
            out = kern.sample(nuts.new_state(dev(q0), tgt), eps, immd, T)
            return out, kern._nuts["holder"]["rng"].clone()
        finally:
            eng.set_option("block_roll", 0)

    eng.set_option("block_dense", 0)
    srng = RandomStream(seeds=seeds)
    kern = nuts.new_kernel(srng, tgt, max_num_expansions=6)
    state, ref, total = nuts.new_state(dev(q0), tgt), [], torch.zeros(C, dtype=torch.int64, device="cuda")
    for _ in range(T):
        info, upd = kern(state, eps, immd)
        ref.append(info)
        total += info.n_leapfrog
        state = info.state._replace(momentum=None)
    ref_rng = upd[srng].clone()
    assert len(set(int(i.n_leapfrog[0]) for i in ref)) > 1 or T < 4  # trees of different lengths: chains do drift apart
    for roll in (0, 1, 2, 16):
        out, rng = sample(roll)
        samples, info, acc, div = out[:4]
        for t in range(T):
            assert torch.equal(samples[t], ref[t].state.position), (roll, t)
            assert torch.equal(acc[t], ref[t].acceptance_probability), (roll, t)
            assert torch.equal(div[t].bool(), ref[t].is_diverging.bool()), (roll, t)
        same_bits(info, ref[-1]._replace(n_leapfrog=info.n_leapfrog))
        assert torch.equal(info.n_leapfrog, total)
        assert torch.equal(rng, ref_rng), roll


@pytest.mark.parametrize("case", ["diverging", "depth1", "one_chain", "per_chain_eps"])
def test_block_dense_rolling_edge_cases_match_lockstep_bitwise(eng, case):
    """The rolling kernel where its scheduling is stressed: chains that diverge at their first step (the scan goes on as a
    phantom while the transition's outputs are final), trees of a single expansion (a chain begins again every other
    round), a workgroup with one chain, one step size per chain (very different tree lengths side by side)."""
    from aehmc_amd import PerChain, RandomStream, nuts
    D, C, T, max_exp, scale = 160, 27, 8, 6, 0.5
    if case == "diverging":
        scale = 40.0      # |dH| > 1000 at the first step of most transitions
    elif case == "depth1":
        max_exp = 1
    elif case == "one_chain":
        C = 1
    r = np.random.default_rng(len(case))
    tgt, _, imm = make("dense", D, r)
    immd, q0 = dev(imm), r.normal(size=(C, D))
    seeds = list(range(40, 40 + C))
    eps = scale * D ** -0.25
    if case == "per_chain_eps":
        eps = PerChain(dev(D ** -0.25 * np.exp(r.uniform(np.log(0.05), np.log(1.5), size=C))))

    def run(roll, single_calls):
        eng.set_option("block_dense", 1 if roll else 0)
        eng.set_option("block_roll", roll)
        try:
            srng = RandomStream(seeds=seeds)
            kern = nuts.new_kernel(srng, tgt, max_num_expansions=max_exp)
            state = nuts.new_state(dev(q0), tgt)
            if not single_calls:
                out = kern.sample(state, eps, immd, T)
                return out[0], out[2], out[3], out[1], kern._nuts["holder"]["rng"].clone()
            pos, acc, div = [], [], []
            for _ in range(T):
                info, upd = kern(state, eps, immd)
                pos.append(info.state.position), acc.append(info.acceptance_probability), div.append(info.is_diverging)
                state = info.state._replace(momentum=None)
            return torch.stack(pos), torch.stack(acc), torch.stack(div), info, upd[srng].clone()
        finally:
            eng.set_option("block_roll", 0)
            eng.set_option("block_dense", 1)

    ref = run(0, True)   # lock-step path, one call per transition
    if case == "diverging":
        assert bool(ref[2].bool().any())
    for roll in (1, 3):
        got = run(roll, False)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2].bool(), ref[2].bool()), roll
        assert torch.equal(got[3].state.potential_energy, ref[3].state.potential_energy)
        assert torch.equal(got[3].state.potential_energy_grad, ref[3].state.potential_energy_grad)
        assert torch.equal(got[4], ref[4]), roll


def test_block_dense_is_independent_of_the_workgroup_a_chain_lands_in(eng):
    """A chain's results do not depend on which chains share its workgroup (rows of the MFMA tile are independent):
    chains 5..12 run alone (one partly filled workgroup) equal the same chains inside a 40-chain call."""
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(21)
    D, C = 150, 40
    tgt, _, imm = make("dense", D, r)
    immd, q0 = dev(imm), r.normal(size=(C, D))
    seeds = list(range(500, 500 + C))

    def run(lo, hi):
        kern = nuts.new_kernel(RandomStream(seeds=seeds[lo:hi]), tgt, max_num_expansions=6)
        info, _ = kern(nuts.new_state(dev(q0[lo:hi]), tgt), 0.25, immd)
        return info

    full, part = run(0, C), run(5, 13)
    assert torch.equal(full.state.position[5:13], part.state.position)
    assert torch.equal(full.n_leapfrog[5:13], part.n_leapfrog)
    assert torch.equal(full.acceptance_probability[5:13], part.acceptance_probability)
