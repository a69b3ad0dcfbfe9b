"""Round-3 additions at the Python / C-ABI boundary: resuming an RNG stream from `updates` (README.md:49-51,
nuts.py:138-153), dual-averaging step-size adaptation as a building block around an HMC kernel
(tests/test_step_size.py:13-88), window adaptation of an HMC kernel, content-keyed parameter caches."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import np_adaptation as na  # noqa: E402


def _case(kind, C=6, D=5):
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(11)
    mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    tgt = targets.DiagGaussian(mu, sigma)
    mod = nuts if kind == "nuts" else hmc
    extra = () if kind == "nuts" else (9,)
    seeds = [300 + c for c in range(C)]
    q0 = torch.as_tensor(r.normal(size=(C, D)), device="cuda")
    return RandomStream, mod, tgt, imm, extra, seeds, q0


@pytest.mark.parametrize("kind", ["nuts", "hmc"])
def test_rng_stream_resumes_from_updates(kind):
    """2 N transitions in one session == N transitions, save (chain state + updates[srng] on the host),
    rebuild the stream with RandomStream.from_state, N more -- bit for bit, generator states included."""
    RandomStream, mod, tgt, imm, extra, seeds, q0 = _case(kind)
    N = 4

    def run(kernel, state, n):
        upd = None
        for _ in range(n):
            info, upd = kernel(state, 0.21, imm, *extra)
            state = info.state._replace(momentum=None)
        return state, upd

    srng = RandomStream(seeds=seeds)
    kernel = mod.new_kernel(srng, tgt)
    ref_state, ref_upd = run(kernel, mod.new_state(q0, tgt), 2 * N)

    srng1 = RandomStream(seeds=seeds)
    k1 = mod.new_kernel(srng1, tgt)
    mid, upd = run(k1, mod.new_state(q0, tgt), N)
    saved_rng = upd[srng1].cpu().numpy().copy()            # what a checkpoint would hold
    saved = [np.asarray(x.cpu()) for x in (mid.position, mid.potential_energy, mid.potential_energy_grad)]
    del k1, srng1, upd

    srng2 = RandomStream.from_state(saved_rng, seeds=seeds)
    k2 = mod.new_kernel(srng2, tgt)
    from aehmc_amd import IntegratorState
    st = IntegratorState(torch.as_tensor(saved[0], device="cuda"), None, torch.as_tensor(saved[1], device="cuda"),
                         torch.as_tensor(saved[2], device="cuda"))
    end, upd2 = run(k2, st, N)
    assert torch.equal(end.position, ref_state.position)
    assert torch.equal(end.potential_energy, ref_state.potential_energy)
    assert torch.equal(upd2[srng2], ref_upd[srng])
    # a second kernel built on the resumed stream gets the call sites an unbroken session would hand out
    assert np.array_equal(srng2.sites(2), srng.sites(2))
    # without the seeds one kernel can resume, a further one is refused
    lone = RandomStream.from_state(saved_rng)
    assert lone.batched and lone.num_chains == len(seeds)
    mod.new_kernel(lone, tgt)
    with pytest.raises(ValueError, match="without its seeds"):
        mod.new_kernel(lone, tgt)
    with pytest.raises(ValueError, match="call sites"):
        (mod.new_kernel)(RandomStream.from_state(saved_rng[:, :1]), tgt)


def test_dual_averaging_update_matches_the_restatement():
    """aehmc_dual_averaging_update against oracle/np_adaptation.dual_averaging_adaptation (itself pinned to
    tests/test_algorithms.py) on random acceptance sequences, non-default gamma / t0 / kappa included."""
    from aehmc_amd.step_size import dual_averaging_adaptation
    r = np.random.default_rng(5)
    for kw in ({}, dict(target_acceptance_rate=0.65, gamma=0.1, t0=5, kappa=0.6)):
        C = 9
        init, update = dual_averaging_adaptation(**kw)
        oinit, oupdate = na.dual_averaging_adaptation(**kw)
        mu = r.random(C) + 0.5
        st = init(mu)
        ost = [oinit(float(m)) for m in mu]
        for _ in range(60):
            p = r.random(C)
            st = update(torch.as_tensor(p, device="cuda"), st)
            ost = [oupdate(float(p[c]), ost[c]) for c in range(C)]
        assert st.step.cpu().tolist() == [int(o.step) for o in ost]
        np.testing.assert_allclose(st.iterates.cpu().numpy(), [o.iterates for o in ost], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(st.iterates_avg.cpu().numpy(), [o.iterates_avg for o in ost], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(st.gradient_avg.cpu().numpy(), [o.gradient_avg for o in ost], rtol=1e-13, atol=1e-15)


def test_dual_averaging_around_an_hmc_kernel():
    """The shape of tests/test_step_size.py:13-88: logprob -2 (x - 1)^2 (= N(1, 1/2)), HMC with 10 integration
    steps, unit metric, start at 1.0, step size exp(x_t) from the dual-averaging state after every transition;
    the mean acceptance probability settles at the 0.8 target and the step size stays in (0.1, 10).  (The
    reference runs one chain for 10 000 steps; here 64 chains x 1500.)"""
    from aehmc_amd import PerChain, RandomStream, hmc, targets
    from aehmc_amd.step_size import dual_averaging_adaptation
    C, n = 64, 1500
    tgt = targets.DiagGaussian(np.array([1.0]), np.array([0.5]))
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(torch.ones(C, 1, dtype=torch.float64, device="cuda"), tgt)
    init, update = dual_averaging_adaptation()
    da = init(torch.ones(C))
    acc = []
    for _ in range(n):
        info, _ = kernel(state, PerChain(torch.exp(da.iterates)), np.ones(1), 10)
        da = update(info.acceptance_probability, da)
        state = info.state._replace(momentum=None)
        acc.append(info.acceptance_probability)
    acc = torch.stack(acc).cpu().numpy()
    assert acc.mean() == pytest.approx(0.8, rel=1e-2)
    eps = torch.exp(da.iterates).cpu().numpy()
    assert (eps < 10).all() and (eps > 1e-1).all()
    assert da.step.cpu().tolist() == [n + 1] * C


def test_window_adaptation_of_an_hmc_kernel():
    """window_adaptation.run drives an HMC kernel when told its trajectory length: per-chain step sizes
    and diagonal mass matrices come back, the warm-up's own acceptance sits at the 0.8 target (sampling with
    the AVERAGED final step size, window_adaptation.py:184-190, accepts somewhat more), and the adapted
    inverse mass matrix tracks the target's variances."""
    from aehmc_amd import RandomStream, hmc, targets, window_adaptation
    C, D = 256, 3
    sigma = np.array([0.5, 1.0, 3.0])
    tgt = targets.DiagGaussian(np.zeros(D), sigma)
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(torch.zeros(C, D, dtype=torch.float64, device="cuda"), tgt)
    with pytest.raises(ValueError, match="num_integration_steps"):
        window_adaptation.run(kernel, state, 50)
    state, (eps, imm), _ = window_adaptation.run(kernel, state, 400, num_integration_steps=8)
    _, info, acc_hist, _ = kernel.sample(state, eps, imm, 8, 200)
    assert 0.75 < acc_hist.mean().item() < 0.99
    e = eps.value.cpu().numpy()
    assert np.isfinite(e).all() and (e > 0.05).all() and (e < 3).all()
    var = imm.value.mean(dim=0).cpu().numpy()
    np.testing.assert_allclose(var, sigma ** 2, rtol=0.35)


@pytest.mark.parametrize("tk,D,C,full", [("diag", 3, 9, False), ("diag", 700, 3, False), ("dense", 12, 5, True),
                                         ("dense", 70, 4, False), ("diag", 1500, 2, False)])
def test_hmc_warmup_in_one_call_equals_step_by_step(tk, D, C, full):
    """aehmc_hmc_warmup (round 3) enqueues the warm-up loop of an HMC kernel -- transition, aehmc_adapt_update,
    transition, ... -- in one C-ABI call: state, step sizes, matrices, the following transition and the generator
    states equal window_adaptation.run(fused=False), the caller-side loop, bit for bit (register-resident, wide,
    small-dense and lock-step HMC paths)."""
    from aehmc_amd import RandomStream, hmc, targets, window_adaptation
    r = np.random.default_rng(D + C)
    if tk == "dense":
        A = r.normal(size=(D, D))
        prec = np.linalg.inv(A @ A.T / D + np.eye(D))
        tgt = targets.DenseMVN(r.normal(size=D), 0.5 * (prec + prec.T))
        q0 = r.normal(size=(C, D))
    else:
        mu, sigma = r.normal(size=D), 0.5 + r.random(D)
        tgt = targets.DiagGaussian(mu, sigma)
        q0 = mu + sigma * r.normal(size=(C, D))
    outs = []
    for fused in (True, False):
        srng = RandomStream(seeds=[800 + c for c in range(C)])
        kernel = hmc.new_kernel(srng, tgt)
        state = hmc.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        state, (eps, imm), upd = window_adaptation.run(kernel, state, 120, is_mass_matrix_full=full, fused=fused,
                                                       num_integration_steps=6)
        assert torch.equal(upd[srng], kernel._hmc["holder"]["rng"])
        info, upd = kernel(state, eps, imm, 6)
        outs.append((state.position.clone(), state.potential_energy.clone(), eps.value.clone(), imm.value.clone(),
                     imm.sqrt_mass.clone(), info.state.position.clone(), info.acceptance_probability.clone(),
                     upd[srng].clone()))
    for k, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), k
    assert torch.isfinite(outs[0][3]).all() and (outs[0][2] > 0).all()


def test_target_parameters_are_keyed_by_content():
    """A numpy parameter edited in place between two calls must be seen (the reference re-reads its graph
    inputs on every call); the same content under another object must not re-upload."""
    from aehmc_amd import hmc, targets
    from aehmc_amd.engine import get_engine
    mu, sigma = np.zeros(4), np.ones(4)
    tgt = targets.DiagGaussian(mu, sigma)
    q = torch.ones(3, 4, dtype=torch.float64, device="cuda")
    U0 = hmc.new_state(q, tgt).potential_energy.clone()
    mu[:] = 1.0  # in place: q == mu now
    U1 = hmc.new_state(q, tgt).potential_energy.clone()
    assert not torch.equal(U0, U1)
    np.testing.assert_allclose(U1.cpu().numpy(), 4 * 0.9189385332046727, rtol=1e-14)
    eng = get_engine()
    kept = eng._keep["target"][1]["mu"].data_ptr()
    hmc.new_state(q, targets.DiagGaussian(np.ones(4), np.ones(4)))  # same content, new objects: cache hit
    assert eng._keep["target"][1]["mu"].data_ptr() == kept


def test_warmup_refuses_copies_of_the_adaptation_state():
    """aehmc_nuts_warmup samples with the bound per-chain arrays while the update kernel rewrites the state's:
    a caller who bound COPIES would warm up with frozen parameters -- every path refuses (C-ABI level)."""
    import ctypes as ct
    from aehmc_amd import PerChain, RandomStream, nuts, targets
    from aehmc_amd.engine import EngineError, get_engine, rng_to_device
    eng = get_engine()
    C, D = 8, 700  # D > 512: the workgroup-per-chain path, i.e. the step-by-step loop inside the library
    tgt = targets.IsoGaussian()
    state = nuts.new_state(torch.zeros(C, D, dtype=torch.float64, device="cuda"), tgt)
    st, cst = eng.adapt_alloc(C, D)
    eng.adapt_init(C, D, 1.0, cst)
    rng = rng_to_device(RandomStream(seeds=list(range(C))).sites(4), eng.device)
    q, U, g = state.position.clone(), state.potential_energy.clone(), state.potential_energy_grad.clone()
    out, c = eng._diag(C, D, True)
    eng.set_metric(PerChain(st["imm"].clone(), st["sqrt_mass"].clone()), D)  # copies
    eng.ensure_workspace(C, 10)
    eng._keep["eps"] = st["step_size"]
    eng._check(eng.lib.aehmc_set_step_sizes(eng.ctx, st["step_size"].data_ptr(), C), "set_step_sizes")
    stage, wend = (ct.c_int32 * 3)(0, 0, 0), (ct.c_int32 * 3)(0, 0, 0)
    with pytest.raises(EngineError, match="adaptation state's own arrays"):
        eng._step_call(eng.lib.aehmc_nuts_warmup, "aehmc_nuts_warmup", eng.ctx, C, rng.data_ptr(), 3, stage, wend, 0.8,
                       10, 1000.0, q.data_ptr(), U.data_ptr(), g.data_ptr(), ct.byref(c), ct.byref(cst), eng.stream)


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)


def test_alternating_dense_metrics_are_factored_once_each():
    """Two kernels with different dense inverse mass matrices alternating on one device: each matrix is uploaded and
    factored (Cholesky + triangular inverse, ~0.1 s at D = 1e4) ONCE -- the engine keeps a handle (device copy + L^-T)
    per metric content -- and the results are those of a fresh engine state.  An in-place edit of a matrix is a new
    content, hence a new factorisation."""
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    r = np.random.default_rng(0)
    D, C = 2000, 8

    def spd():
        A = r.normal(size=(D, D))
        M = A @ A.T / D + np.eye(D)
        return 0.5 * (M + M.T)

    imms = [spd(), spd()]
    tgt = targets.StdNormal()
    q0 = r.normal(size=(C, D))
    kerns = [nuts.new_kernel(RandomStream(seeds=list(range(10 * k, 10 * k + C))), tgt, max_num_expansions=3) for k in range(2)]
    states = [nuts.new_state(dev(q0), tgt) for _ in range(2)]
    n0 = eng.n_metric_factorizations
    first = []
    for rep in range(3):
        for k in range(2):
            info, _ = kerns[k](states[k], 0.05, imms[k])
            states[k] = info.state._replace(momentum=None)
            if rep == 0:
                first.append(info.state.position.clone())
    assert eng.n_metric_factorizations - n0 == 2
    # same transitions from a cold cache (force): same bits
    k2 = nuts.new_kernel(RandomStream(seeds=list(range(0, C))), tgt, max_num_expansions=3)
    eng.set_metric(imms[0], D, force=True)
    info, _ = k2(nuts.new_state(dev(q0), tgt), 0.05, imms[0])
    assert torch.equal(info.state.position, first[0])
    n1 = eng.n_metric_factorizations
    imms[1][5, 5] += 0.5  # in-place edit: new content
    kerns[1](states[1], 0.05, imms[1])
    assert eng.n_metric_factorizations == n1 + 1
