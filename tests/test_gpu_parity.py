"""Parity of the HIP path (through the C-ABI) with the CPU oracle on identical seeds.

Tolerance: the north star asks for 1e-6 relative against the reference's CPU path on
identical RNG seeds; these tests use RTOL = 1e-9 (observed ~1e-13: only reduction order
differs) and require every discrete output (num_doublings, flags, leapfrog counts, RNG
consumption) to be identical.  D = 1 cases have no reductions and must be bit-exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle as co  # noqa: E402

RTOL = 1e-9


@pytest.fixture(scope="module")
def eng():
    from aehmc_amd.engine import get_engine
    return get_engine()


def dev(x, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda").to(dtype)


# ------------------------------------------------------------------ RNG
def test_device_rng_matches_numpy(eng):
    from aehmc_amd.engine import rng_to_device
    seeds = [0, 1, 59, 2**40 + 7, 123456789]
    st = co.site_states(seeds, 2)
    n = 100_000
    rng = rng_to_device(st[:, 0].copy(), "cuda")
    z = eng.rng_normals(rng, n).cpu().numpy()
    for c, s in enumerate(seeds):
        ch = np.random.SeedSequence(s).spawn(2)
        g = np.random.default_rng(ch[0])
        ref = g.normal(0, 1, size=n)
        same = z[c] == ref
        # the tail / wedge branches call log1p/exp: allow 1-ulp there, nowhere else
        assert same.mean() > 0.999
        np.testing.assert_allclose(z[c], ref, rtol=4e-16, atol=0)
        # generator state after n draws identical (same number of raw draws consumed)
        after = rng.cpu().numpy().view(np.uint64)[c]
        stt = g.bit_generator.state["state"]["state"]
        assert int(after[0]) == stt >> 64 and int(after[1]) == stt & (2**64 - 1)
    # request sizes around the 64-lane round: every way a call can end (inside an accepted run,
    # on a resolved rejection, on a restart), continuing the same streams
    gens = [np.random.default_rng(np.random.SeedSequence(s).spawn(2)[0]) for s in seeds]
    for g in gens:
        g.normal(0, 1, size=n)
    for m in list(range(1, 70)) + [127, 128, 129, 191, 1000, 4097] * 3:
        zz = eng.rng_normals(rng, m).cpu().numpy()
        for c, g in enumerate(gens):
            np.testing.assert_allclose(zz[c], g.normal(0, 1, size=m), rtol=4e-16, atol=0)
    after = rng.cpu().numpy().view(np.uint64)
    for c, g in enumerate(gens):
        stt = g.bit_generator.state["state"]["state"]
        assert int(after[c][0]) == stt >> 64 and int(after[c][1]) == stt & (2**64 - 1)
    ps = np.random.default_rng(7).random((len(seeds), 20_000))
    ps[:, ::7], ps[:, ::11], ps[:, ::13] = 0.0, 1.0, 0.5
    rng2 = rng_to_device(st[:, 1].copy(), "cuda")
    b = eng.rng_bernoulli(rng2, dev(ps)).cpu().numpy()
    for c, s in enumerate(seeds):
        g = np.random.default_rng(np.random.SeedSequence(s).spawn(2)[1])
        ref = np.array([g.binomial(1, p) for p in ps[c]])
        assert np.array_equal(b[c], ref)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (128, 128, 16), (130, 257, 50), (64, 300, 1000),
                                   (517, 129, 333)])
def test_gemm_f64_mfma(eng, M, N, K):
    r = np.random.default_rng(M * 1000 + N)
    A, B = r.normal(size=(M, K)), r.normal(size=(N, K))  # asymmetric operands
    out = eng.gemm_nt(dev(A), dev(B)).cpu().numpy()
    np.testing.assert_allclose(out, A @ B.T, rtol=1e-12, atol=1e-12 * np.sqrt(K))


@pytest.mark.timeout(120)
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("M,N,K", [(1100, 8000, 200), (1408, 6000, 96), (2900, 2560, 1000), (4096, 1300, 50),
                                   (3000, 9000, 40)])
def test_gemm_streamk_bitwise_equals_tiled(eng, M, N, K, mode):
    """Stream-K splits tiles between neighbouring workgroups but continues the same k-chain
    from the published partial accumulators: bit-identical to the one-tile-per-workgroup kernel,
    in the whole-tile rounds, in the split tail and in the strided mode alike (mode 2: the
    128 x 256-tile variant).  Also with a compacted (gathered) row list."""
    r = np.random.default_rng(M + N + K)
    A, B = dev(r.normal(size=(M, K))), dev(r.normal(size=(N, K)))
    eng.set_option("streamk", mode)
    out1 = eng.gemm_nt(A, B)
    out1b = eng.gemm_nt(A, B)   # second launch: new epoch, flags of the first one are stale
    eng.set_option("streamk", 0)
    out0 = eng.gemm_nt(A, B)
    eng.set_option("streamk", 2)
    assert torch.equal(out1, out0) and torch.equal(out1b, out0)
    np.testing.assert_allclose(out1.cpu().numpy(), A.cpu().numpy() @ B.cpu().numpy().T, rtol=1e-12,
                               atol=1e-12 * np.sqrt(K))


@pytest.mark.parametrize("M", [1, 8, 16, 17, 32, 33, 64, 65, 100, 128])
@pytest.mark.parametrize("N,K", [(300, 1000), (10000, 512), (16, 16), (1001, 38)])
def test_gemm_tail_kernel_bitwise_equals_tiled(eng, M, N, K):
    """Up to 128 rows take the bandwidth-bound tail kernel (one wave per 16 columns): the same rows
    inside a taller product (tiled kernels) give the same bits; also through a gathered row list."""
    r = np.random.default_rng(M * 1000 + N + K)
    A, B = dev(r.normal(size=(M + 200, K))), dev(r.normal(size=(N, K)))
    small = eng.gemm_nt(A[:M].contiguous(), B)
    tall = eng.gemm_nt(A, B)
    assert torch.equal(small, tall[:M])
    np.testing.assert_allclose(small.cpu().numpy(), A[:M].cpu().numpy() @ B.cpu().numpy().T, rtol=1e-12,
                               atol=1e-12 * np.sqrt(K))


@pytest.mark.parametrize("M", [129, 200, 256, 257, 300, 384, 385, 500])
@pytest.mark.parametrize("N,K", [(10000, 512), (1000, 64), (1300, 10000)])
def test_gemm_few_row_tiles_every_tile_written(eng, M, N, K):
    """129 .. ~768 rows take the pipelined 128 x 128 kernel on a persistent grid; with fewer tiles than
    workgroups every workgroup owns at most one tile and the tile count (2 or 3 row tiles x 79 / 8 / 11
    column tiles) is NOT a multiple of the 8 XCDs -- a round-1 bug left the last slot of the higher
    XCDs unvisited there (stale output for a few 128 x 128 blocks, c3's live-row counts 129 .. 384).
    The output buffer is poisoned first; result vs torch and bitwise vs the one-tile-per-workgroup kernel."""
    r = np.random.default_rng(M + N + K)
    A, B = dev(r.normal(size=(M, K))), dev(r.normal(size=(N, K)))
    ref = (A @ B.T)
    outs = {}
    for mode in (2, 1, 0):
        eng.set_option("streamk", mode)
        poison = torch.full((M, N), float("nan"), dtype=torch.float64, device="cuda")
        eng.lib.aehmc_gemm_nt(eng.ctx, M, N, K, A.data_ptr(), K, B.data_ptr(), K, poison.data_ptr(), N, eng.stream)
        torch.cuda.synchronize()
        outs[mode] = poison
    eng.set_option("streamk", 2)
    for mode, out in outs.items():
        assert torch.isfinite(out).all(), (mode, (~torch.isfinite(out)).sum().item())
        assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12 * np.sqrt(K)), mode
    assert torch.equal(outs[2], outs[0]) and torch.equal(outs[1], outs[0])


@pytest.mark.parametrize("M,N,K", [(4096, 200, 200), (700, 300, 300), (129, 65, 65), (1000, 201, 201), (333, 1000, 77),
                                   (1500, 1100, 130), (130, 2048, 33), (2000, 96, 1000)])
def test_gemm_small_tiles_bitwise_equals_tiled(eng, M, N, K):
    """Mid-size products (fewer than 256 tiles of 128 x 128, N <= 2048) take (32 I) x (32 J) tiles so that every CU
    has work (round 3: D = 200, 4096 chains 37 -> 17 us): 64 x 128, 64 x 64 and 32 x 64, chosen by tile count or
    forced with the option; every variant gives the bits of the 128 x 128 kernel (same K-tiles, same k order), writes
    every element (poisoned output), odd leading dimensions (scalar loads) and K ending inside a K-tile included."""
    r = np.random.default_rng(M + N + K)
    A, B = dev(r.normal(size=(M, K))), dev(r.normal(size=(N, K)))
    outs = {}
    try:
        for mode in (0, 1, 2, 3, 4):
            eng.set_option("gemm_small_tiles", mode)
            poison = torch.full((M, N), float("nan"), dtype=torch.float64, device="cuda")
            eng.lib.aehmc_gemm_nt(eng.ctx, M, N, K, A.data_ptr(), K, B.data_ptr(), K, poison.data_ptr(), N, eng.stream)
            torch.cuda.synchronize()
            outs[mode] = poison
    finally:
        eng.set_option("gemm_small_tiles", 1)
    assert torch.allclose(outs[0], A @ B.T, rtol=1e-12, atol=1e-12 * np.sqrt(K))
    for mode in (1, 2, 3, 4):
        assert torch.equal(outs[mode], outs[0]), mode


@pytest.mark.parametrize("sampler,D,C", [("nuts", 100, 300), ("hmc", 200, 1000), ("nuts", 333, 160)])
def test_mid_size_dense_transitions_do_not_depend_on_the_gemm_tiles(eng, sampler, D, C):
    """The lock-step path of a mid-size dense problem with its products on the small tiles (default) and on the
    128 x 128 / tail kernels (`gemm_small_tiles` = 0): the same transitions bit for bit -- positions, energies,
    diagnostics, generator states -- because every GEMM variant sums each output element's k-chain in the same order."""
    from aehmc_amd import RandomStream, hmc, nuts
    r = np.random.default_rng(D + C)
    tgt, _, imm = make_case("dense", "dense", D, r)
    q0 = r.normal(size=(C, D))
    mod, extra = (nuts, ()) if sampler == "nuts" else (hmc, (12,))
    outs = []
    try:
        for mode in (1, 0):
            eng.set_option("gemm_small_tiles", mode)
            srng = RandomStream(seeds=[60 + c for c in range(C)])
            kernel = mod.new_kernel(srng, tgt)
            state = mod.new_state(dev(q0), tgt)
            for _ in range(2):
                info, upd = kernel(state, 0.3 / D ** 0.25, imm, *extra)
                state = info.state._replace(momentum=None)
            outs.append((info.state.position.clone(), info.state.potential_energy.clone(), info.state.momentum.clone(),
                         info.acceptance_probability.clone(), info.n_leapfrog.clone(), upd[srng].clone()))
    finally:
        eng.set_option("gemm_small_tiles", 1)
    for k, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), k


# ------------------------------------------------------------------ G1 on the GPU
def test_g1_readme_bit_exact_on_gpu():
    """README.md:22-54 through the drop-in API: position after one NUTS transition."""
    from aehmc_amd import RandomStream, nuts, targets
    srng = RandomStream(seed=0)
    target = targets.StdNormal()
    kernel = nuts.new_kernel(srng, target)
    state = nuts.new_state(0.0, target)
    info, updates = kernel(state, 1e-2, 1.0)
    assert info.state.position.item() == 1.1034719409361107
    assert info.num_doublings.item() == 8 and info.n_leapfrog.item() == 136
    assert not info.is_diverging.item() and not info.is_turning.item()
    assert info.acceptance_probability.item() == pytest.approx(0.9999767760191554, rel=1e-14)
    assert srng in updates


# ------------------------------------------------------------------ G2 / G3 on the GPU
def test_g2_g3_regression_on_gpu(regression_data):
    """examples/LinearRegression.ipynb: log-density at [3, log 10] (:188) and the single HMC
    step from [3, log .21], eps=5e-5, L=1024, imm=[1,1], seed 0 (:293-297)."""
    from aehmc_amd import RandomStream, hmc, targets
    X, y = regression_data
    tgt = targets.LinearRegression(X, y)
    s = hmc.new_state(np.array([3.0, np.log(10.0)]), tgt)
    assert -s.potential_energy.item() == pytest.approx(-32238.026021294307, rel=1e-12)
    kernel = hmc.new_kernel(RandomStream(seed=0), tgt)
    info, _ = kernel(hmc.new_state(np.array([3.0, np.log(0.21)]), tgt), 5e-5, np.array([1.0, 1.0]), 1024)
    np.testing.assert_allclose(info.state.position.cpu().numpy(), [2.99946192, -1.30494977], atol=5e-9)
    assert info.state.potential_energy.item() == pytest.approx(12433.00653542, abs=5e-9)
    np.testing.assert_allclose(info.state.potential_energy_grad.cpu().numpy(),
                               [-489.93218536, -22571.36970197], atol=5e-9)
    assert info.acceptance_probability.item() == 1.0 and not info.is_diverging.item()


@pytest.mark.parametrize("N,C,L,mk", [(10_000, 1, 64, "diag"), (10_000, 6, 33, "diag"), (10_176, 9, 20, "diag"),
                                      (10_177, 5, 20, "diag"), (25_001, 7, 12, "diag"), (700, 4, 50, "diag"),
                                      (10_000, 1, 64, "dense"), (10_177, 5, 20, "dense"), (700, 6, 50, "dense")])
def test_regression_hmc_fused_matches_oracle(eng, N, C, L, mk):
    """HMC on the regression target in one launch (hmc_linreg.cuh): rows resident in LDS (N <= 10176)
    or streamed by direct loads (N above), workgroups with 1..4 live chains, three consecutive
    transitions through kernel.sample -- against the oracle, and against the lock-step path
    (`fused_hmc` = 0), which differs in the order of the row sums only (dense 2 x 2 metric, round 3: and in the
    products, an MFMA GEMM there)."""
    from aehmc_amd import PerChain, RandomStream, hmc, targets
    r = np.random.default_rng(N + C)
    X = r.normal(0, 1, size=N)
    y = 3 * X + 0.5 * r.normal(0, 1, size=N)  # (row-wise noise: a well-conditioned posterior for any N)
    tgt, otgt = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y)
    imm = np.array([0.25 / N, 0.5 / N])  # ~ posterior variances of w and log n
    if mk == "dense":
        imm = np.array([[0.25 / N, -0.08 / N], [-0.08 / N, 0.5 / N]])
    eps = 0.3
    seeds = [500 + c for c in range(C)]
    q0 = np.array([3.0, np.log(0.5)]) + (0.5 / np.sqrt(N)) * r.normal(size=(C, 2))
    metric = co.Metric(imm, 2)
    outs = {}
    for fused in (1, 0):
        eng.set_option("fused_hmc", fused)
        kernel = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
        state = hmc.new_state(dev(q0), tgt)
        samples, info, acc, div = kernel.sample(state, eps, imm, L, 3)
        outs[fused] = (samples.cpu().numpy(), info, acc.cpu().numpy())
    eng.set_option("fused_hmc", 1)
    rng = co.site_states(seeds, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    for t in range(3):
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        for fused in (1, 0):
            np.testing.assert_allclose(outs[fused][0][t], q, rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(outs[fused][2][t], res["acceptance_probability"], rtol=1e-7, atol=1e-12)
    for fused in (1, 0):
        info = outs[fused][1]
        np.testing.assert_allclose(info.state.potential_energy.cpu().numpy(), U, rtol=RTOL)
        np.testing.assert_allclose(info.state.potential_energy_grad.cpu().numpy(), g, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(info.state.momentum.cpu().numpy(), res["momentum"], rtol=1e-7, atol=1e-10)
        assert np.array_equal(info.is_diverging.cpu().numpy(), res["is_diverging"])
        assert (info.n_leapfrog == 3 * L).all()
    assert 0.2 < outs[1][2].mean() <= 1.0  # a real trajectory, not a frozen chain


@pytest.mark.parametrize("metric_kind,C,resident", [("diag", 12, 2), ("diag", 6, 2), ("diag", 5, 0), ("dense", 12, 2)])
def test_regression_nuts_matches_oracle(eng, regression_data, metric_kind, C, resident):
    """NUTS on the notebook's regression posterior (notebook cell 36 settings).  Diagonal metric:
    the workgroup-cooperative resident kernel (4 chains per workgroup; C = 6 leaves a workgroup
    half empty) or, with resident_nuts=0, the lock-step kernels; dense metric: lock-step."""
    from aehmc_amd import RandomStream, nuts, targets
    X, y = regression_data
    eng.set_option("resident_nuts", resident)
    r = np.random.default_rng(2)
    tgt, otgt = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y)
    imm = np.array([2.13e-05, 4.43e-05])
    if metric_kind == "dense":
        imm = np.array([[2.13e-05, 5e-6], [5e-6, 4.43e-05]])
    eps = 0.8
    seeds = [77 + c for c in range(C)]
    q0 = np.array([3.0, np.log(0.49)]) + 0.01 * r.normal(size=(C, 2))
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt)
    state = nuts.new_state(dev(q0), tgt)
    rng, metric = co.site_states(seeds, 4), co.Metric(imm, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    for _ in range(3):
        info, _ = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g)
        # sums over 1e4 rows in a different order: 1e-9 on the state, exact discrete outputs
        check_state(info, q, U, g, res)
        state = info.state._replace(momentum=None)
    eng.set_option("resident_nuts", 2)


@pytest.mark.parametrize("N,metric_kind,max_exp", [(37, "diag", 10), (10177, "scalar", 10), (30001, "diag", 10),
                                                  (30001, "diag", 2), (12288, "diag", 10), (10177, "dense", 10),
                                                  (30001, "dense", 10)])
def test_regression_nuts_row_counts_match_oracle(eng, N, metric_kind, max_exp):
    """k_nuts_linreg over the row-count regimes of its sweep: fewer rows than threads (37), all rows in
    LDS (<= 10176), LDS rows + streamed blocks + an odd tail (30001), LDS rows + whole blocks only
    (12288 = 10176 + 2112: one 512-piece block per wave short of 8 waves), a scalar metric, a dense 2 x 2 metric
    and a tree cut by max_num_expansions.  Row-wise noise so that the posterior has a real width."""
    from aehmc_amd import RandomStream, nuts, targets
    r = np.random.default_rng(N)
    X = r.normal(size=N)
    y = 3 * X + 0.5 * r.normal(size=N)
    tgt, otgt = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y)
    C = 5
    imm = np.float64(1.0 / N) if metric_kind == "scalar" else np.array([1.0 / N, 0.5 / N])
    if metric_kind == "dense":  # k_nuts_linreg<DM> (round 3): a full 2 x 2 inverse mass matrix
        imm = np.array([[1.0 / N, 0.22 / N], [0.22 / N, 0.5 / N]])
    eps = 0.5
    seeds = [31 + c for c in range(C)]
    q0 = np.array([3.0, np.log(0.5)]) + 0.02 * r.normal(size=(C, 2))
    kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    rng = co.site_states(seeds, 4)
    metric = co.Metric(imm, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    lengths = []
    for _ in range(3):
        info, _ = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp)
        check_state(info, q, U, g, res)
        lengths.append(res["n_leapfrog"])
        state = info.state._replace(momentum=None)
    assert np.max(lengths) >= 3 and 0.1 < info.acceptance_probability.mean().item() <= 1.0  # real trajectories


@pytest.mark.parametrize("family", ["nuts-linreg", "nuts-linreg-sample", "hmc-linreg", "hmc-fused-d1", "nuts-team1-sample",
                                    "nuts-lockstep-d1", "nuts-wave-d1"])
def test_rng_state_after_many_momentum_draws(eng, regression_data, family):
    """Every kernel family that draws its momentum itself, with enough chains x transitions that the
    ziggurat's redraw path (1.5 % of normals, a data-dependent loop on the generator state) is taken
    dozens of times at small D: the per-chain generator states after the run equal numpy's / the
    oracle's, and so do the discrete outputs.  (Round 2: with `unsigned __int128` state the compiler
    lost the upper state word inside that loop in ONE instantiation -- k_nuts_resident<1,1> -- which no
    test reached; rng.cuh now keeps explicit 64-bit halves.)"""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    r = np.random.default_rng(len(family))
    # (D = 1 families: ~10^4 normals, so that the ziggurat's TAIL loop -- 0.03 % of draws -- is taken too)
    C, T = (512 if "linreg" in family else 3072), 4
    seeds = [8000 + c for c in range(C)]
    if "linreg" in family:
        X, y = regression_data
        tgt, otgt, D = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y), 2
        imm, eps = np.array([2.13e-05, 4.43e-05]), 0.8
        q0 = np.array([3.0, np.log(0.49)]) + 0.01 * r.normal(size=(C, 2))
    else:
        D = 1
        mu, sigma, imm, eps = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D), 0.2
        tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
        q0 = r.normal(size=(C, D))
    metric = co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    is_hmc = family.startswith("hmc")
    rng = co.site_states(seeds, 2 if is_hmc else 4)
    srng = RandomStream(seeds=seeds)
    try:
        if family == "nuts-team1-sample":
            eng.set_option("resident_min_team", 1)
        if family == "nuts-lockstep-d1":
            eng.set_option("resident_nuts", 0)
        if family == "nuts-wave-d1":
            eng.set_option("resident_nuts", 1)
        if is_hmc:
            kernel = hmc.new_kernel(srng, tgt)
            state = hmc.new_state(dev(q0), tgt)
            _, info, _, _ = kernel.sample(state, eps, imm, 7, T)
            for _ in range(T):
                res = co.hmc_step(otgt, metric, rng, eps, 7, q, U, g)
            rng_dev = kernel._hmc["holder"]["rng"] if hasattr(kernel, "_hmc") else None
        else:
            kernel = nuts.new_kernel(srng, tgt)
            state = nuts.new_state(dev(q0), tgt)
            if family.endswith("sample"):
                _, info, _, _ = kernel.sample(state, eps, imm, T)
            else:
                for _ in range(T):
                    info, _ = kernel(state, eps, imm)
                    state = info.state._replace(momentum=None)
            for _ in range(T):
                res = co.nuts_step(otgt, metric, rng, eps, q, U, g)
            rng_dev = kernel._nuts["holder"]["rng"]
            assert np.array_equal(info.num_doublings.cpu().numpy().reshape(-1), res["num_doublings"])
    finally:
        eng.set_option("resident_min_team", 0)
        eng.set_option("resident_nuts", 2)
    if rng_dev is not None:
        got = rng_dev.cpu().numpy().view(np.uint64).reshape(rng.shape)
        bad = np.nonzero((got[:, :, :2] != rng[:, :, :2]).any(axis=(1, 2)))[0]
        assert bad.size == 0, f"generator state differs for chains {bad.tolist()[:10]}"
    np.testing.assert_allclose(info.state.position.cpu().numpy().reshape(q.shape), q, rtol=1e-7, atol=1e-10)
    assert np.array_equal(info.is_diverging.cpu().numpy().reshape(-1), res["is_diverging"])


@pytest.mark.parametrize("D,sampler", [(200, "nuts"), (64, "nuts"), (200, "hmc"), (700, "nuts")])
def test_dense_path_equals_diagonal_path_under_triangular_map(D, sampler):
    """The product's dense path (dense-MVN target + dense inverse mass matrix: fp64 MFMA GEMMs, L^-T
    momentum, lock-step engine) against its diagonal path (resident kernels) through the invariance of
    tests/test_oracle_golden.py::test_dense_branch_equals_diagonal_branch_under_triangular_map: the image
    of a diagonal problem under a lower-triangular map q' = A q has the same tree shapes, RNG
    consumption and acceptance, and q'_t = A q_t.  Both dense modes (`dense_linear` 1 / 0)."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    r = np.random.default_rng(D)
    mu, sigma, m = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    A = np.diag(0.7 + 0.6 * r.random(D)) + 0.3 * np.tril(r.normal(size=(D, D)), -1) / np.sqrt(D)
    Ainv = np.linalg.inv(A)
    P = Ainv.T @ np.diag(1 / sigma ** 2) @ Ainv
    imm = A @ np.diag(m) @ A.T
    P, imm = 0.5 * (P + P.T), 0.5 * (imm + imm.T)
    C, eps = 9, 0.3 / D ** 0.25
    seeds = [90 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    mod = nuts if sampler == "nuts" else hmc
    extra = () if sampler == "nuts" else (11,)

    def run(tgt, metric, start, mode):
        eng.set_option("dense_linear", mode)
        srng = RandomStream(seeds=seeds)
        kernel = mod.new_kernel(srng, tgt)
        state = mod.new_state(dev(start), tgt)
        outs = []
        for _ in range(3):
            info, upd = kernel(state, eps, metric, *extra)
            state = info.state._replace(momentum=None)
            outs.append(info)
        return outs, upd[srng].cpu().numpy().copy()

    try:
        diag, rng_d = run(targets.DiagGaussian(mu, sigma), m, q0, 1)
        for mode in (1, 0):
            dense, rng_m = run(targets.DenseMVN(A @ mu, P), imm, q0 @ A.T, mode)
            for a, b in zip(dense, diag):
                assert torch.equal(a.n_leapfrog, b.n_leapfrog) and torch.equal(a.is_diverging, b.is_diverging)
                if sampler == "nuts":
                    assert torch.equal(a.num_doublings, b.num_doublings) and torch.equal(a.is_turning, b.is_turning)
                np.testing.assert_allclose(a.state.position.cpu().numpy(), b.state.position.cpu().numpy() @ A.T,
                                           rtol=1e-9, atol=1e-10)
                np.testing.assert_allclose(a.state.momentum.cpu().numpy(), b.state.momentum.cpu().numpy() @ Ainv,
                                           rtol=1e-8, atol=1e-9)
                np.testing.assert_allclose(a.acceptance_probability.cpu().numpy(),
                                           b.acceptance_probability.cpu().numpy(), rtol=1e-9)
            assert np.array_equal(rng_m, rng_d)
    finally:
        eng.set_option("dense_linear", 1)


@pytest.mark.parametrize("case", ["teams-1", "teams-64", "wide", "linreg"])
def test_resident_kernels_with_diverging_trajectories_match_oracle(eng, regression_data, case):
    """The single-launch NUTS kernels where trajectories diverge (|H0 - E| > threshold) -- on the first
    leapfrog of an expansion (trajectory.py:336: the tuple of the initial state is returned, the scan
    still runs and draws its uniforms) and later: flags, leapfrog counts, values and the generator
    states of all four call sites against the oracle."""
    from aehmc_amd import RandomStream, nuts, targets
    r = np.random.default_rng(len(case) + 3)
    thr = 5.0
    if case == "linreg":
        X, y = regression_data
        D, C = 2, 10
        tgt, otgt = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y)
        imm, eps = np.array([2.13e-05, 4.43e-05]), 40.0
        q0 = np.array([3.0, np.log(0.49)]) + 0.01 * r.normal(size=(C, 2))
    else:
        D, C = {"teams-1": (3, 50), "teams-64": (20, 6), "wide": (700, 4)}[case]
        mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
        tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
        eps = 3.5 / D ** 0.25 if D < 100 else 2.5 / D ** 0.25
        q0 = r.normal(size=(C, D))
    seeds = [20 + c for c in range(C)]
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt, divergence_threshold=thr)
    state = nuts.new_state(dev(q0), tgt)
    rng, metric = co.site_states(seeds, 4), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    eng.set_option("resident_min_team", 1 if case == "teams-1" else 0)
    ndiv = first = 0
    try:
        for _ in range(4):
            info, updates = kernel(state, eps, imm)
            res = co.nuts_step(otgt, metric, rng, eps, q, U, g, thr=thr)
            check_state(info, q, U, g, res)
            assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
            state = info.state._replace(momentum=None)
            ndiv += int(res["is_diverging"].sum())
            first += int(((res["n_leapfrog"] == 1) & (res["is_diverging"] != 0)).sum())
    finally:
        eng.set_option("resident_min_team", 0)
    assert ndiv >= 3 and first >= 1, (ndiv, first)


# ------------------------------------------------------------------ helpers
def make_case(kind, tkind, D, r):
    from aehmc_amd import targets
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    if tkind == "dense":
        A = r.normal(size=(D, D))
        cov = A @ A.T / D + np.eye(D)
        prec = np.linalg.inv(cov)
        prec = 0.5 * (prec + prec.T)
        tgt, otgt = targets.DenseMVN(mu, prec), co.Target(co.T_DENSE_MVN, D, mu=mu, prec=prec)
    elif tkind == "diag":
        tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    elif tkind == "std":
        tgt, otgt = targets.StdNormal(), co.Target(co.T_STD_NORMAL, D)
    else:
        tgt, otgt = targets.IsoGaussian(), co.Target(co.T_ISO_GAUSSIAN, D)
    if kind == "scalar":
        imm = np.float64(0.7)
    elif kind == "diag":
        imm = 0.5 + r.random(D)
    else:
        A = r.normal(size=(D, D))
        imm = A @ A.T / D + np.eye(D)
        imm = 0.5 * (imm + imm.T)
    return tgt, otgt, imm


def check_state(info, q, U, g, res, nuts=True):
    np.testing.assert_allclose(info.state.position.cpu().numpy().reshape(q.shape), q, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.state.potential_energy.cpu().numpy().reshape(U.shape), U, rtol=RTOL)
    np.testing.assert_allclose(info.state.potential_energy_grad.cpu().numpy().reshape(g.shape), g,
                               rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.state.momentum.cpu().numpy().reshape(q.shape), res["momentum"],
                               rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(info.acceptance_probability.cpu().numpy().reshape(-1),
                               res["acceptance_probability"], rtol=RTOL)
    assert np.array_equal(info.is_diverging.cpu().numpy().reshape(-1), res["is_diverging"])
    assert np.array_equal(info.n_leapfrog.cpu().numpy().reshape(-1), res["n_leapfrog"])
    if nuts:
        assert np.array_equal(info.num_doublings.cpu().numpy().reshape(-1), res["num_doublings"])
        assert np.array_equal(info.is_turning.cpu().numpy().reshape(-1), res["is_turning"])


CASES = [("scalar", "std", 1), ("diag", "std", 3), ("diag", "diag", 70), ("diag", "iso", 200),
         ("dense", "diag", 33), ("diag", "dense", 33), ("dense", "dense", 150), ("dense", "dense", 64),
         ("dense", "diag", 65), ("dense", "dense", 700)]


@pytest.mark.parametrize("kind,tkind,D", CASES)
def test_nuts_matches_oracle(kind, tkind, D):
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(D * 7 + len(kind))
    tgt, otgt, imm = make_case(kind, tkind, D, r)
    C, eps, max_exp = 6, 0.25 if D < 100 else 0.12, 6
    seeds = [100 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    rng = co.site_states(seeds, 4)
    metric = co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    np.testing.assert_allclose(state.potential_energy.cpu().numpy(), U, rtol=1e-12)
    for _ in range(4):
        info, updates = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp)
        check_state(info, q, U, g, res)
        # RNG consumption identical at all four call sites
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)


@pytest.mark.parametrize("kind,tkind,D", CASES)
def test_hmc_matches_oracle(kind, tkind, D):
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D * 11 + len(tkind))
    tgt, otgt, imm = make_case(kind, tkind, D, r)
    C, eps, L = 5, 0.2 if D < 100 else 0.1, 9
    seeds = [500 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    srng = RandomStream(seeds=seeds)
    kernel = hmc.new_kernel(srng, tgt)
    state = hmc.new_state(dev(q0), tgt)
    rng = co.site_states(seeds, 2)
    metric = co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    for _ in range(4):
        info, updates = kernel(state, eps, imm, L)
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        check_state(info, q, U, g, res, nuts=False)
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)


@pytest.mark.parametrize("linear,compact", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_dense_options_match_oracle(eng, linear, compact):
    """literal (3 GEMMs/leapfrog, as metrics.py forms imm.p) and linear (velocity carried by
    linearity, 2 GEMMs/leapfrog) dense paths, with and without live-chain compaction."""
    from aehmc_amd import RandomStream, hmc, nuts
    eng.set_option("dense_linear", linear)
    eng.set_option("compact", compact)
    try:
        D, C, eps, max_exp = 150, 9, 0.12, 7
        r = np.random.default_rng(77)
        tgt, otgt, imm = make_case("dense", "dense", D, r)
        seeds = [900 + c for c in range(C)]
        q0 = r.normal(size=(C, D))
        srng = RandomStream(seeds=seeds)
        kernel = nuts.new_kernel(srng, tgt, max_num_expansions=max_exp)
        hkernel = hmc.new_kernel(srng, tgt)
        state = nuts.new_state(dev(q0), tgt)
        rng, hrng = co.site_states(seeds, 4), co.site_states(seeds, 2, first_site=4)
        metric = co.Metric(imm, D)
        q, U, g = co.new_state(otgt, q0.copy())
        for _ in range(3):
            info, _ = kernel(state, eps, imm)
            res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp)
            check_state(info, q, U, g, res)
            state = info.state._replace(momentum=None)
        info, _ = hkernel(state, eps, imm, 11)   # dense HMC continues the same srng (sites 5, 6)
        res = co.hmc_step(otgt, metric, hrng, eps, 11, q, U, g)
        check_state(info, q, U, g, res, nuts=False)
    finally:
        eng.set_option("dense_linear", 1)
        eng.set_option("compact", 1)


def test_dense_metric_not_positive_definite_is_an_error():
    """aehmc_set_metric factors the dense inverse mass matrix itself (blocked Cholesky on the
    fp64 GEMM); a non-PD matrix is reported, not silently used."""
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import EngineError
    tgt = targets.StdNormal()
    state = nuts.new_state(dev(np.zeros((2, 3))), tgt)
    kernel = nuts.new_kernel(RandomStream(seeds=[0, 1]), tgt)
    bad = np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]])  # symmetric, indefinite
    with pytest.raises(EngineError, match="positive definite"):
        kernel(state, 0.1, bad)


@pytest.mark.parametrize("tk", ["std", "iso", "diag"])
@pytest.mark.parametrize("D", [40, 100, 200, 400, 900, 1500, 3000, 6000, 9000])
def test_hmc_every_fused_instantiation_matches_oracle(D, tk):
    """One case per compiled variant of the single-launch HMC kernels: k_hmc_fused<R> for R = 1, 2, 4, 8,
    16 elements per lane (D = 40 ... 900) and k_hmc_wide<256,8>, <512,8>, <1024,8>, <1024,10> (D = 1500
    ... 9000), each for the three coordinate-wise targets -- three transitions through sample(), values,
    acceptance, divergence and the generator state against the oracle.  (A compiler problem in one
    instantiation of one kernel went unnoticed in round 1 because no test reached it.)"""
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D + len(tk))
    tgt, otgt, imm = make_case("diag", tk, D, r)
    C, L, T = 5, 7, 3
    seeds = [1500 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.25 / D ** 0.25
    metric, rng = co.Metric(imm, D), co.site_states(seeds, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    kernel = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    samples, info, acc, div = kernel.sample(hmc.new_state(dev(q0), tgt), eps, imm, L, T)
    for t in range(T):
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        np.testing.assert_allclose(samples[t].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(acc[t].cpu().numpy(), res["acceptance_probability"], rtol=RTOL)
        assert np.array_equal(div[t].cpu().numpy(), res["is_diverging"])
    check_state(info._replace(n_leapfrog=info.n_leapfrog // T), q, U, g, res, nuts=False)
    got = kernel._hmc["holder"]["rng"].cpu().numpy().view(np.uint64).reshape(rng.shape)
    assert np.array_equal(got[:, :, :2], rng[:, :, :2])
    assert acc.mean().item() > 0.5


@pytest.mark.parametrize("D,C,tk", [(1500, 5, "diag"), (3000, 3, "std"), (5000, 3, "iso"), (10000, 3, "diag")])
def test_hmc_resident_large_d_matches_oracle(eng, D, C, tk):
    """D > 1024: momentum pre-pass (one wavefront per chain) + one workgroup per chain with the
    state in registers for all L steps.
    Against the oracle (1e-9) and against the lock-step path (1e-12), incl. sample()."""
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D)
    tgt, otgt, imm = make_case("diag", tk, D, r)
    seeds = [800 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps, L = 0.2 / D ** 0.25, 11
    metric = co.Metric(imm, D)
    rng = co.site_states(seeds, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    kernel = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    state = hmc.new_state(dev(q0), tgt)
    for _ in range(2):
        info, upd = kernel(state, eps, imm, L)
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        check_state(info, q, U, g, res, nuts=False)
        state = info.state._replace(momentum=None)
    # multi-transition calls, two back to back, then a single step: the RNG streams continue
    samples, info2, acc, div = kernel.sample(state, eps, imm, L, 3)
    for t_ in range(3):
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        np.testing.assert_allclose(samples[t_].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
    samples, info2, acc, div = kernel.sample(info2.state._replace(momentum=None), eps, imm, L, 4)
    for t_ in range(4):
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        np.testing.assert_allclose(samples[t_].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(acc[t_].cpu().numpy(), res["acceptance_probability"], rtol=RTOL)
    info2, _ = kernel(info2.state._replace(momentum=None), eps, imm, L)
    res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
    check_state(info2, q, U, g, res, nuts=False)
    eng.set_option("fused_hmc", 0)  # lock-step path on the same seeds
    k2 = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    s2 = hmc.new_state(dev(q0), tgt)
    for _ in range(10):
        i2, _ = k2(s2, eps, imm, L)
        s2 = i2.state._replace(momentum=None)
    eng.set_option("fused_hmc", 1)
    np.testing.assert_allclose(info2.state.position.cpu().numpy(), i2.state.position.cpu().numpy(),
                               rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("D,tk", [(5000, "iso"), (10000, "std"), (8190, "iso")])
def test_hmc_wide_many_chains_equal_small_call(eng, D, tk):
    """More chains than CUs on the workgroup-per-chain HMC kernel (several rounds of workgroups per CU):
    chains of a 700-chain call equal, bit for bit, the same chains (same seeds) run in a small call,
    over two transitions of sample() -- results do not depend on what shares the launch."""
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D)
    tgt, otgt, imm = make_case("diag", tk, D, r)
    C, L = 700, 6
    seeds = [3000 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.2 / D ** 0.25
    kernel = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
    samples, info, acc, div = kernel.sample(hmc.new_state(dev(q0), tgt), eps, imm, L, 2)
    sel = [0, 1, 255, 256, 257, 511, 512, 699]
    k2 = hmc.new_kernel(RandomStream(seeds=[seeds[i] for i in sel]), tgt)
    s2, i2, a2, d2 = k2.sample(hmc.new_state(dev(q0[sel]), tgt), eps, imm, L, 2)
    assert torch.equal(samples[:, sel], s2) and torch.equal(acc[:, sel], a2)
    assert torch.equal(info.state.momentum[sel], i2.state.momentum)
    assert torch.equal(info.state.potential_energy[sel], i2.state.potential_energy)
    # ... and the oracle for the first of them
    metric, rng = co.Metric(imm, D), co.site_states(seeds[:2], 2)
    q, U, g = co.new_state(otgt, q0[:2].copy())
    for t_ in range(2):
        co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        np.testing.assert_allclose(samples[t_, :2].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
    assert 0.3 < acc.mean().item() <= 1.0


@pytest.mark.parametrize("D,tk", [(1500, "diag"), (9000, "iso"), (10000, "diag")])
def test_hmc_wide_long_sample_with_rejections(eng, D, tk):
    """The round-3 workgroup-per-chain HMC kernel keeps the position on chip for the transitions of a launch, parks
    the state a rejection falls back to in LDS and takes the momenta of up to 22 transitions per launch pair:
    50 transitions in one sample() call (three chunks) at a step size that rejects about every third proposal equal,
    bit for bit, 50 single-transition calls on the same seeds -- positions after every transition, acceptance
    history, final state, last momentum and the generator states -- and the oracle along the way (1e-9)."""
    from aehmc_amd import RandomStream, hmc
    r = np.random.default_rng(D + 7)
    tgt, otgt, imm = make_case("diag", tk, D, r)
    C, L, T = 3, 5, 50
    seeds = [4100 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 1.35 / D ** 0.25
    k1, s1 = hmc.new_kernel(RandomStream(seeds=seeds), tgt), RandomStream(seeds=seeds)
    samples, info, acc, div = k1.sample(hmc.new_state(dev(q0), tgt), eps, imm, L, T)
    k2 = hmc.new_kernel(s1, tgt)
    state = hmc.new_state(dev(q0), tgt)
    metric, rng = co.Metric(imm, D), co.site_states(seeds, 2)
    q, U, g = co.new_state(otgt, q0.copy())
    n_acc = 0
    for t in range(T):
        i2, upd = k2(state, eps, imm, L)
        state = i2.state._replace(momentum=None)
        assert torch.equal(samples[t], i2.state.position), t
        assert torch.equal(acc[t], i2.acceptance_probability), t
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        np.testing.assert_allclose(samples[t].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
        n_acc += int(res["accepted"].sum())
    assert torch.equal(info.state.position, i2.state.position)
    assert torch.equal(info.state.potential_energy, i2.state.potential_energy)
    assert torch.equal(info.state.potential_energy_grad, i2.state.potential_energy_grad)
    assert torch.equal(info.state.momentum, i2.state.momentum)
    assert torch.equal(k1._hmc["holder"]["rng"], upd[s1])
    assert 0.1 * C * T < n_acc < 0.95 * C * T  # both branches of the accept decision, many times


@pytest.mark.parametrize("sampler", ["nuts", "hmc"])
@pytest.mark.parametrize("kind,tkind", [("dense", "dense"), ("dense", "diag"), ("dense", "iso"), ("diag", "dense"), ("scalar", "dense")])
@pytest.mark.parametrize("D", [1, 7, 64])
def test_small_dense_single_launch_kernels(eng, sampler, kind, tkind, D):
    """k_nuts_resident<64,1,.,DENSE> / k_hmc_fused_dense (round 3: dense metric and / or dense target, D <= 64, the whole
    transition in one launch with the products inside the wavefront), one case per compiled variant and sampler at
    D = 1, 7 and the maximum 64 (three matrices = 96 KB of LDS), 13 chains (a partial workgroup), three
    transitions: against the oracle (values 1e-9, every discrete output and the generator states exact) and
    against the lock-step path with its MFMA GEMMs (1e-11: another summation order, same arithmetic otherwise)."""
    from aehmc_amd import RandomStream, hmc, nuts
    r = np.random.default_rng(1000 * D + len(kind) * 7 + len(tkind))
    tgt, otgt, imm = make_case(kind, tkind, D, r)
    C = 13
    seeds = [77 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.35 / D ** 0.25
    mod = nuts if sampler == "nuts" else hmc
    extra = () if sampler == "nuts" else (9,)
    metric, rng = co.Metric(imm, D), co.site_states(seeds, 4 if sampler == "nuts" else 2)
    q, U, g = co.new_state(otgt, q0.copy())
    finals = []
    for fused in (1, 0):
        eng.set_option("resident_nuts", 2 if fused else 0)
        eng.set_option("fused_hmc", fused)
        try:
            srng = RandomStream(seeds=seeds)
            kernel = mod.new_kernel(srng, tgt) if sampler == "hmc" else mod.new_kernel(srng, tgt, max_num_expansions=7)
            state = mod.new_state(dev(q0), tgt)
            for t in range(3):
                info, upd = kernel(state, eps, imm, *extra)
                state = info.state._replace(momentum=None)
                if fused:
                    res = (co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=7) if sampler == "nuts"
                           else co.hmc_step(otgt, metric, rng, eps, 9, q, U, g))
                    check_state(info, q, U, g, res, nuts=sampler == "nuts")
            if fused:
                assert np.array_equal(upd[srng].cpu().numpy().view(np.uint64).reshape(rng.shape), rng)
            finals.append((info.state.position.clone(), info.n_leapfrog.clone(), upd[srng].clone()))
        finally:
            eng.set_option("resident_nuts", 2)
            eng.set_option("fused_hmc", 1)
    np.testing.assert_allclose(finals[0][0].cpu().numpy(), finals[1][0].cpu().numpy(), rtol=1e-11, atol=1e-13)
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])


@pytest.mark.parametrize("sampler", ["nuts", "hmc"])
@pytest.mark.parametrize("per_chain", [False, True])
@pytest.mark.parametrize("tkind,D", [("dense", 11), ("diag", 64), ("dense", 64)])
def test_small_dense_sample_equals_single_calls(eng, sampler, per_chain, tkind, D):
    """kernel.sample(T) of a small dense problem (NUTS: all T transitions inside one launch of the MULTI instantiation,
    generator states in registers in between) returns bit for bit what T single calls return -- positions, histories,
    the last transition's diagnostics and the generator states; shared and per-chain dense matrices."""
    from aehmc_amd import PerChain, RandomStream, hmc, nuts
    r = np.random.default_rng(31 * D + len(tkind))
    tgt, _, imm = make_case("dense", tkind, D, r)
    C, T = 11, 6
    if per_chain:
        A = r.normal(size=(C, D, D))
        imm = A @ A.transpose(0, 2, 1) / D + 0.5 * np.eye(D)
        imm = PerChain(dev(0.5 * (imm + imm.transpose(0, 2, 1))))
        eps = PerChain(dev(0.4 / D ** 0.25 * (0.5 + r.random(C))))
    else:
        eps = 0.4 / D ** 0.25
    seeds = [900 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    mod = nuts if sampler == "nuts" else hmc
    extra = () if sampler == "nuts" else (5,)
    s1, s2 = RandomStream(seeds=seeds), RandomStream(seeds=seeds)
    k1, k2 = mod.new_kernel(s1, tgt), mod.new_kernel(s2, tgt)
    samples, info, acc, div = k1.sample(mod.new_state(dev(q0), tgt), eps, imm, *extra, T)
    state = mod.new_state(dev(q0), tgt)
    nleap = 0
    for t in range(T):
        i2, upd = k2(state, eps, imm, *extra)
        state = i2.state._replace(momentum=None)
        nleap = nleap + i2.n_leapfrog
        assert torch.equal(samples[t], i2.state.position), t
        assert torch.equal(acc[t], i2.acceptance_probability), t
        assert torch.equal(div[t].bool(), i2.is_diverging.bool()), t
    assert torch.equal(info.state.potential_energy, i2.state.potential_energy)
    assert torch.equal(info.state.potential_energy_grad, i2.state.potential_energy_grad)
    assert torch.equal(info.state.momentum, i2.state.momentum)
    if sampler == "nuts":
        assert torch.equal(info.n_leapfrog, nleap)  # sample(): the total over the T transitions
        assert torch.equal(info.num_doublings, i2.num_doublings)
    holder = k1._nuts["holder"] if sampler == "nuts" else k1._hmc["holder"]
    assert torch.equal(holder["rng"], upd[s2])


def test_hmc_fused_equals_lockstep_bitwise(eng):
    """The register-resident single-launch HMC kernel and the generic lock-step path run
    the same arithmetic in the same order."""
    from aehmc_amd import RandomStream, hmc, targets
    r = np.random.default_rng(0)
    D, C, L = 100, 64, 32
    q0 = r.normal(size=(C, D))
    imm = 0.5 + r.random(D)
    outs = []
    for fused in (1, 0):
        eng.set_option("fused_hmc", fused)
        tgt = targets.DiagGaussian(r.normal(size=D) * 0 + 0.3, np.full(D, 1.7))
        srng = RandomStream(seeds=list(range(C)))
        kernel = hmc.new_kernel(srng, tgt)
        state = hmc.new_state(dev(q0), tgt)
        for _ in range(3):
            info, _ = kernel(state, 0.1, imm, L)
            state = info.state._replace(momentum=None)
        outs.append((info.state.position.cpu().numpy(), info.state.momentum.cpu().numpy(),
                     info.acceptance_probability.cpu().numpy()))
    eng.set_option("fused_hmc", 1)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("D,C", [(1, 5), (130, 40), (300, 6), (512, 5)])
def test_nuts_resident_equals_lockstep_bitwise(eng, D, C):
    """Register-resident single-launch NUTS with one wavefront per chain (128 < D <= 512; and
    D = 1, which has no reduction) runs the lock-step path's arithmetic in the same order:
    identical bits, identical RNG use."""
    from aehmc_amd import RandomStream, nuts, targets
    r = np.random.default_rng(D)
    q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    outs = []
    for resident in (1, 0):
        eng.set_option("resident_nuts", resident)
        tgt = targets.DiagGaussian(mu, sigma)
        srng = RandomStream(seeds=list(range(C)))
        kernel = nuts.new_kernel(srng, tgt, max_num_expansions=7)
        state = nuts.new_state(dev(q0), tgt)
        for _ in range(3):
            info, upd = kernel(state, 0.25 / D ** 0.25, imm)
            state = info.state._replace(momentum=None)
        outs.append((info.state.position.cpu().numpy(), info.state.momentum.cpu().numpy(),
                     info.state.potential_energy.cpu().numpy(), info.state.potential_energy_grad.cpu().numpy(),
                     info.acceptance_probability.cpu().numpy(), info.n_leapfrog.cpu().numpy(),
                     info.num_doublings.cpu().numpy(), info.is_turning.cpu().numpy(), upd[srng].cpu().numpy()))
    eng.set_option("resident_nuts", 2)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("D,C", [(1, 300), (2, 70), (3, 33), (7, 50), (16, 9), (24, 21), (40, 19), (100, 13), (128, 5),
                                 (700, 4), (2048, 3), (3000, 3), (10000, 2)])
def test_nuts_resident_teams_match_oracle(eng, D, C):
    """Sub-wavefront teams (D <= 128: 64/T chains per wave, divergent SIMT control flow) and
    workgroup teams (D > 512: cross-wave reductions): summation order differs from the oracle
    and from the lock-step path -> 1e-9 on values, exact discrete outputs and RNG use."""
    from aehmc_amd import RandomStream, nuts, targets
    eng.set_option("resident_min_team", 1)
    r = np.random.default_rng(D + 1)
    mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
    tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    metric, seeds = co.Metric(imm, D), [70 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.25 / D ** 0.25
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt, max_num_expansions=6)
    state = nuts.new_state(dev(q0), tgt)
    rng = co.site_states(seeds, 4)
    q, U, g = co.new_state(otgt, q0.copy())
    for _ in range(3):
        info, updates = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=6)
        check_state(info, q, U, g, res)
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)
    eng.set_option("resident_min_team", 0)


@pytest.mark.parametrize("D,tk", [(600, "diag"), (2500, "iso"), (5000, "iso"), (9000, "std"), (8200, "diag")])
def test_nuts_wide_deep_trees_match_oracle(D, tk):
    """The workgroup-per-chain kernel on DEEP trees (step size 20x too small: 8-9 doublings, sub-trajectories
    of up to 257 steps, U-turn checks over up to 8 checkpoint levels): everything the kernel does
    differently from the literal algorithm is exercised at depth -- the first check level taken from
    registers (incl. the stale step-0 indices of every later expansion), deeper levels prefetched or
    fetched on demand, checkpoint pairs that are never read back not stored, trajectory ends parked
    only on a change of direction, the initial state aliased instead of copied.  Every variant:
    registers only (D = 600, 2500), q in LDS with imm beside it (5000, 9000), q and dU/dq in LDS
    with streamed parameters (8200, diagonal-Gaussian target)."""
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(D)
    tgt, otgt, imm = make_case("diag", tk, D, r)
    C = 3
    seeds = [40 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.012 / D ** 0.25
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt, max_num_expansions=9)
    state = nuts.new_state(dev(q0), tgt)
    rng, metric = co.site_states(seeds, 4), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    deepest = 0
    for _ in range(3):
        info, updates = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=9)
        check_state(info, q, U, g, res)
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)
        deepest = max(deepest, int(res["num_doublings"].max()))
    assert deepest >= 8, deepest


@pytest.mark.parametrize("tk", ["std", "iso", "diag"])
@pytest.mark.parametrize("D", [700, 1500, 3000, 6000, 9000])
def test_nuts_every_wide_instantiation_matches_oracle(D, tk):
    """One case per compiled variant of the workgroup-per-chain NUTS kernel: k_nuts_wide<256,4>, <256,8>,
    <512,8>, <512,16,q in LDS>, <512,20,q in LDS> (D = 700 ... 9000) x the three coordinate-wise
    targets; trees of 4-6 doublings, two transitions, values, discrete outputs and generator state."""
    from aehmc_amd import RandomStream, nuts
    r = np.random.default_rng(3 * D + len(tk))
    tgt, otgt, imm = make_case("diag", tk, D, r)
    C = 3
    seeds = [60 + c for c in range(C)]
    q0 = r.normal(size=(C, D))
    eps = 0.05 / D ** 0.25
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt)
    state = nuts.new_state(dev(q0), tgt)
    rng, metric = co.site_states(seeds, 4), co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    for _ in range(2):
        info, updates = kernel(state, eps, imm)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g)
        check_state(info, q, U, g, res)
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)
    assert res["num_doublings"].max() >= 4


def test_nuts_fused_equals_lockstep_bitwise(eng):
    """The single-launch NUTS kernel (one wavefront loops a chain's whole tree) and the
    one-launch-per-leapfrog lock-step path share their device functions: same bits."""
    from aehmc_amd import RandomStream, nuts, targets
    r = np.random.default_rng(1)
    D, C = 130, 40
    q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    outs = []
    for fused in (1, 0):
        eng.set_option("fused_nuts", fused)
        tgt = targets.DiagGaussian(mu, sigma)
        srng = RandomStream(seeds=list(range(C)))
        kernel = nuts.new_kernel(srng, tgt, max_num_expansions=7)
        state = nuts.new_state(dev(q0), tgt)
        for _ in range(3):
            info, upd = kernel(state, 0.2, imm)
            state = info.state._replace(momentum=None)
        outs.append((info.state.position.cpu().numpy(), info.state.momentum.cpu().numpy(),
                     info.acceptance_probability.cpu().numpy(), info.n_leapfrog.cpu().numpy(),
                     info.num_doublings.cpu().numpy(), upd[srng].cpu().numpy()))
    eng.set_option("fused_nuts", 0)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("fused", [1, 0])
def test_hmc_sample_equals_repeated_steps(eng, fused):
    """kernel.sample(N) (one launch on the fused path) == N calls of kernel(...)"""
    from aehmc_amd import RandomStream, hmc, targets
    r = np.random.default_rng(4)
    D, C, L, N = 70, 16, 7, 6
    q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
    tgt = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
    eng.set_option("fused_hmc", fused)
    try:
        k1 = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        k2 = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        s1 = hmc.new_state(dev(q0), tgt)
        samples, info, acc_hist, div_hist = k1.sample(s1, 0.15, imm, L, N)
        s2 = hmc.new_state(dev(q0), tgt)
        for t in range(N):
            i2, _ = k2(s2, 0.15, imm, L)
            s2 = i2.state._replace(momentum=None)
            assert torch.equal(samples[t], i2.state.position)
            assert torch.equal(acc_hist[t], i2.acceptance_probability)
        assert torch.equal(info.state.position, i2.state.position)
        assert torch.equal(info.state.momentum, i2.state.momentum)
        assert torch.equal(info.state.potential_energy, i2.state.potential_energy)
        assert info.n_leapfrog.sum().item() == C * L * N
    finally:
        eng.set_option("fused_hmc", 1)


# ------------------------------------------------------------------ divergence / phantom scan
@pytest.mark.parametrize("step_size, div, turn, doublings",
                         [(100000.0, True, False, 1), (0.0000001, False, False, 10), (1.0, False, True, 1)])
def test_multiplicative_expansion_outcomes_on_gpu(step_size, div, turn, doublings):
    # /root/reference tests/test_trajectory.py:144-208 (U = x^2/2, q = 1, imm = 1.0, seed 59)
    from aehmc_amd import RandomStream, nuts, targets
    tgt = targets.IsoGaussian()
    kernel = nuts.new_kernel(RandomStream(seed=59), tgt)
    info, _ = kernel(nuts.new_state(1.0, tgt), step_size, 1.0)
    assert (info.is_diverging.item(), info.is_turning.item(), info.num_doublings.item()) == \
        (div, turn, doublings)
    if doublings == 10:
        assert info.n_leapfrog.item() == 1033


def test_divergent_first_step_keeps_rng_in_step_with_oracle():
    """trajectory.py:336: when the first step of a sub-trajectory diverges the scan still
    runs and consumes RNG; a following transition must still agree with the oracle."""
    from aehmc_amd import RandomStream, nuts, targets
    D, C = 4, 8
    r = np.random.default_rng(9)
    tgt, otgt = targets.StdNormal(), co.Target(co.T_STD_NORMAL, D)
    seeds = list(range(40, 40 + C))
    q0 = r.normal(size=(C, D)) * 3
    srng = RandomStream(seeds=seeds)
    kernel = nuts.new_kernel(srng, tgt)
    state = nuts.new_state(dev(q0), tgt)
    rng = co.site_states(seeds, 4)
    imm = np.ones(D)
    metric = co.Metric(imm, D)
    q, U, g = co.new_state(otgt, q0.copy())
    for eps in (30.0, 0.3, 45.0, 0.2):  # 30/45: |delta| > 1000 on the very first leapfrog
        info, updates = kernel(state, eps, imm)
        with np.errstate(all="ignore"):
            res = co.nuts_step(otgt, metric, rng, eps, q, U, g)
        check_state(info, q, U, g, res)
        assert np.array_equal(updates[srng].cpu().numpy().view(np.uint64)[:, :, :2], rng[:, :, :2])
        state = info.state._replace(momentum=None)
    assert res is not None


# ------------------------------------------------------------------ known-answer tables via the C-ABI
def test_kinetic_energy_and_turning_tables(eng):
    from aehmc_amd import targets
    # tests/test_metrics.py:39-68, 71-120
    for imm, p, expected in [(np.float64(1.0), [1.0], 0.5), (np.ones(1), [1.0], 0.5),
                             (np.ones(2), [1.0, 1.0], 1.0), (np.eye(2), [1.0, 1.0], 1.0)]:
        D = len(p)
        eng.set_target(targets.IsoGaussian(), D)
        eng.set_metric(imm, D)
        assert eng.kinetic_energy(dev([p])).item() == expected
        ones = dev(np.ones((1, D)))
        assert eng.is_turning(ones, ones, ones).item() is True
    with pytest.raises(ValueError):  # tests/test_metrics.py:123-127
        eng.set_metric(np.ones((2, 2, 2)), 2)


def test_velocity_verlet_analytic(eng):
    from aehmc_amd import targets
    # tests/test_integrators.py:58-67: harmonic oscillator, 100 steps of 0.01
    eng.set_target(targets.IsoGaussian(), 1)
    eng.set_metric(np.ones(1), 1)
    q, p = dev([[0.0]]), dev([[1.0]])
    U, g = eng.new_state(q)
    eng.leapfrog(0.01, 100, q, p, U, g)
    assert q.item() == pytest.approx(np.sin(1.0), abs=1e-2)
    assert p.item() == pytest.approx(np.cos(1.0), abs=1e-2)
    assert U.item() + 0.5 * p.item() ** 2 == pytest.approx(0.5, rel=1e-4)
    # and bit for bit the oracle's trajectory
    t, m = co.Target(co.T_ISO_GAUSSIAN, 1), co.Metric(np.ones(1), 1)
    qo, Uo, go = co.new_state(t, [[0.0]])
    po = np.array([[1.0]])
    co.leapfrog(t, m, 0.01, 100, qo, po, Uo, go)
    assert q.item() == qo[0, 0] and p.item() == po[0, 0]


# ------------------------------------------------------------------ full-size config c2
def test_config2_full_size_properties_and_subset_parity():
    """BASELINE config 2: D=100 isotropic Gaussian, HMC L=32, 4096 chains.
    All chains: acceptance sane, energy error small; first 32 chains: equal to the oracle."""
    from aehmc_amd import RandomStream, hmc, targets
    C, D, L, eps = 4096, 100, 32, 0.1
    q0 = np.random.default_rng(1234).standard_normal((C, D))
    seeds = [1000 + c for c in range(C)]
    tgt = targets.IsoGaussian()
    srng = RandomStream(seeds=seeds)
    kernel = hmc.new_kernel(srng, tgt)
    state = hmc.new_state(dev(q0), tgt)
    imm = np.ones(D)
    n_sub = 32
    otgt, metric = co.Target(co.T_ISO_GAUSSIAN, D), co.Metric(imm, D)
    rng = co.site_states(seeds[:n_sub], 2)
    q, U, g = co.new_state(otgt, q0[:n_sub].copy())
    for _ in range(3):
        info, _ = kernel(state, eps, imm, L)
        res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        state = info.state._replace(momentum=None)
        np.testing.assert_allclose(info.state.position[:n_sub].cpu().numpy(), q, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(info.acceptance_probability[:n_sub].cpu().numpy(),
                                   res["acceptance_probability"], rtol=RTOL)
    acc = info.acceptance_probability.cpu().numpy()
    assert acc.mean() > 0.9 and not info.is_diverging.any().item()
    assert info.n_leapfrog.sum().item() == C * L
    # stationarity: positions stay ~N(0, I)
    assert abs(info.state.position.var().item() - 1.0) < 0.02


# ------------------------------------------------------------------ full-size config c5 (one GPU's shard)
def test_config5_regression_warmup_properties():
    """BASELINE config 5 at one GPU's share: regression scaled to 1e5 rows (notebook generator),
    D=2, 1024 of the 8192 chains, NUTS + window adaptation.  Size-independent properties:
    every chain finds the posterior mode (w ~ 3, n ~ |noise|), adapted step sizes are finite
    and positive, leapfrog counts are consistent, and the state equals a fresh new_state."""
    from aehmc_amd import RandomStream, nuts, targets, window_adaptation
    rng = np.random.default_rng(0)
    N, C = 100_000, 1024
    X = rng.normal(0, 1, size=(N,))
    y = 3 * X + rng.normal(0, 1)
    tgt = targets.LinearRegression(X, y)
    q0 = np.array([3.0, np.log(0.5)]) + 0.05 * rng.normal(size=(C, 2))
    kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), tgt)
    state = nuts.new_state(dev(q0), tgt)
    last, (eps, imm), _ = window_adaptation.run(kernel, state, 40)
    e, m = eps.value.cpu().numpy(), imm.value.cpu().numpy()
    assert np.isfinite(e).all() and (e > 0).all() and np.isfinite(m).all() and (m > 0).all()
    info, _ = kernel(last, eps, imm)
    pos = info.state.position.cpu().numpy()
    assert np.isfinite(pos).all()
    assert abs(np.median(pos[:, 0]) - 3.0) < 0.05
    # (the reference's warm-up starts at step size exp(0) = 1, far too large for this posterior:
    #  most early transitions diverge on their first leapfrog -- n_leapfrog == 1 is legitimate)
    assert (info.n_leapfrog.cpu().numpy() >= 1).all() and np.median(e) < 1.0
    fresh = nuts.new_state(info.state.position, tgt)
    np.testing.assert_allclose(fresh.potential_energy.cpu().numpy(),
                               info.state.potential_energy.cpu().numpy(), rtol=1e-10)


# ------------------------------------------------------------------ full-size config c3
@pytest.mark.timeout(600)
def test_config3_full_size_dense_nuts():
    """BASELINE config 3 at full size (D=1e4 dense precision + dense mass, 4096 chains), tree
    depth capped at 3 so that the oracle can follow: chains 0 and 4095 equal the oracle; for
    all chains the returned (U, grad) equal a fresh evaluation at the returned position, the
    leapfrog count matches the number of expansions, and energy is conserved."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_c3
    from aehmc_amd import RandomStream, nuts, targets
    C, D, max_exp = 4096, 10_000, 3
    Sigma, P = build_c3(D, torch.device("cuda"))
    mu = torch.zeros(D, dtype=torch.float64, device="cuda")
    tgt = targets.DenseMVN(mu, P)
    eps = 0.5 * D ** -0.25
    seeds = [1000 + c for c in range(C)]
    q0 = np.random.default_rng(1234).standard_normal((C, D))
    kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    info, _ = kernel(state, eps, Sigma)
    nd, nl = info.num_doublings.cpu().numpy(), info.n_leapfrog.cpu().numpy()
    full = np.cumsum([2 ** j + 1 for j in range(max_exp)])
    assert ((nd >= 1) & (nd <= max_exp)).all()
    assert (nl <= full[nd - 1]).all() and (nl > np.concatenate([[0], full])[nd - 1]).all()
    assert info.acceptance_probability.mean().item() > 0.9 and not info.is_diverging.any().item()
    fresh = nuts.new_state(info.state.position, tgt)
    np.testing.assert_allclose(fresh.potential_energy.cpu().numpy(), info.state.potential_energy.cpu().numpy(),
                               rtol=1e-11)
    np.testing.assert_allclose(fresh.potential_energy_grad[:64].cpu().numpy(),
                               info.state.potential_energy_grad[:64].cpu().numpy(), rtol=1e-9, atol=1e-9)
    # two chains against the oracle
    sel = [0, C - 1]
    otgt = co.Target(co.T_DENSE_MVN, D, mu=np.zeros(D), prec=P.cpu().numpy())
    metric = co.Metric(Sigma.cpu().numpy(), D)
    rng = co.site_states([seeds[i] for i in sel], 4)
    q, U, g = co.new_state(otgt, q0[sel].copy())
    res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp, nthreads=2)
    np.testing.assert_allclose(info.state.position[sel].cpu().numpy(), q, rtol=RTOL, atol=1e-11)
    assert nl[sel].tolist() == res["n_leapfrog"].tolist() and nd[sel].tolist() == res["num_doublings"].tolist()
    np.testing.assert_allclose(info.acceptance_probability[sel].cpu().numpy(), res["acceptance_probability"],
                               rtol=RTOL)


@pytest.mark.timeout(600)
def test_dense_mid_size_wide_gemm_with_compaction_matches_oracle():
    """2048 chains x D=4096, dense precision and dense mass: the chain-batched products run on the
    software-pipelined 128x256 kernel (16 x 16 tiles), chains finish at different depths, so the
    live-row list shrinks through whole-tile rounds, split tails and partial tile rows.  Three
    chains against the oracle over two transitions."""
    from aehmc_amd import RandomStream, nuts, targets
    C, D, max_exp = 2048, 4096, 5
    r = np.random.default_rng(77)
    i = np.arange(D)
    sig = 1.0 + (i % 3)
    band = (0.4 ** np.abs(np.subtract.outer(np.arange(64), np.arange(64))))  # AR(1) blocks of 64
    Sigma = np.zeros((D, D))
    for b in range(D // 64):
        s = slice(64 * b, 64 * b + 64)
        Sigma[s, s] = band * np.outer(sig[s], sig[s])
    P = np.linalg.inv(Sigma)
    P = 0.5 * (P + P.T)
    mu = r.normal(size=D)
    tgt, otgt = targets.DenseMVN(mu, P), co.Target(co.T_DENSE_MVN, D, mu=mu, prec=P)
    eps = 0.25  # U-turns after ~13 steps: trees end at depth 3..5
    seeds = [3000 + c for c in range(C)]
    q0 = mu + r.normal(size=(C, D)) * sig
    kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
    state = nuts.new_state(dev(q0), tgt)
    sel = [0, 777, C - 1]
    metric = co.Metric(Sigma, D)
    rng = co.site_states([seeds[k] for k in sel], 4)
    q, U, g = co.new_state(otgt, q0[sel].copy())
    depths = []
    for _ in range(2):
        info, _ = kernel(state, eps, dev(Sigma))
        state = info.state._replace(momentum=None)
        res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp, nthreads=3)
        np.testing.assert_allclose(info.state.position[sel].cpu().numpy(), q, rtol=RTOL, atol=1e-10)
        assert info.n_leapfrog[sel].cpu().tolist() == res["n_leapfrog"].tolist()
        assert info.num_doublings[sel].cpu().tolist() == res["num_doublings"].tolist()
        depths.append(info.num_doublings.cpu().numpy())
    assert len(np.unique(np.concatenate(depths))) >= 2  # chains did finish at different depths
