#!/usr/bin/env python3
"""Randomised parity sweep: random (sampler, target, metric, D, C, step size, tree depth, engine options)
against the C oracle -- every discrete output and the generator state exact, values to 1e-9.  A net
for variants that no hand-written test reaches.  usage: fuzz_parity.py [seconds] [seed]   (FUZZ_COUNT=N or FUZZ_CASES=a,b,... : fixed case lists)"""
import os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aehmc_amd import PerChain, RandomStream, hmc, nuts, targets
from aehmc_amd.engine import get_engine
from oracle import c_oracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine()
RTOL = 1e-9


def one(case):
    r = np.random.default_rng(case)
    sampler = r.choice(["nuts", "hmc"])
    tk = r.choice(["std", "iso", "diag", "dense", "linreg"], p=[0.15, 0.15, 0.4, 0.15, 0.15])
    mk = r.choice(["scalar", "diag", "dense"], p=[0.15, 0.6, 0.25])
    if tk == "linreg":
        D, mk = 2, ("diag" if mk == "scalar" else mk)
    elif tk == "dense" or mk == "dense":
        D = int(r.choice([2, 5, 33, 64, 65, 130, 300, 520, 1030], p=[0.15, 0.15, 0.15, 0.1, 0.1, 0.15, 0.1, 0.05, 0.05]))
    else:
        D = int(r.choice([1, 2, 3, 5, 8, 17, 31, 64, 65, 100, 128, 129, 257, 512, 513, 700, 1100, 2100, 4100, 6000]))
    if mk == "scalar" and D > 1 and tk != "linreg":
        mk = "diag"
    C = int(r.choice([1, 2, 3, 5, 9, 17, 33, 70]))
    if D > 2000:
        C = min(C, 5)
    opts = {"resident_nuts": int(r.choice([0, 1, 2])), "resident_min_team": int(r.integers(0, 2)),
            "fused_hmc": int(r.integers(0, 2)), "dense_linear": int(r.integers(0, 2)), "fused_nuts": int(r.integers(0, 2)),
            "streamk": int(r.choice([0, 1, 2])), "compact": int(r.integers(0, 2)),
            "block_dense": int(r.choice([0, 1, 2], p=[0.2, 0.6, 0.2])), "block_roll": int(r.choice([0, 1, 2, 5, 16]))}
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    if tk == "linreg":
        N = int(r.choice([37, 1000, 10176, 10177, 23001]))
        X = r.normal(size=N); y = 3 * X + 0.5 * r.normal(size=N)
        tgt, otgt = targets.LinearRegression(X, y), co.Target(co.T_LINREG, 2, X=X, y=y)
        q0 = np.array([3.0, np.log(0.5)]) + 0.02 * r.normal(size=(C, 2))
        imm = np.array([1.0 / N, 0.5 / N]) if mk == "diag" else np.array([[1.0 / N, 0.1 / N], [0.1 / N, 0.5 / N]])
        eps = 0.5 * float(r.choice([1.0, 0.3, 3.0]))
    else:
        q0 = r.normal(size=(C, D))
        if tk == "dense":
            A = r.normal(size=(D, D)); cov = A @ A.T / D + np.eye(D); prec = np.linalg.inv(cov); prec = 0.5 * (prec + prec.T)
            tgt, otgt = targets.DenseMVN(mu, prec), co.Target(co.T_DENSE_MVN, D, mu=mu, prec=prec)
        elif tk == "diag":
            tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
        elif tk == "std":
            tgt, otgt = targets.StdNormal(), co.Target(co.T_STD_NORMAL, D)
        else:
            tgt, otgt = targets.IsoGaussian(), co.Target(co.T_ISO_GAUSSIAN, D)
        if mk == "dense":
            B = r.normal(size=(D, D)); imm = B @ B.T / D + np.eye(D); imm = 0.5 * (imm + imm.T)
        elif mk == "diag":
            imm = 0.5 + r.random(D)
        else:
            imm = np.float64(0.5 + r.random())
        eps = float(r.choice([0.25, 0.08, 1.5])) / D ** 0.25
    thr = float(r.choice([1000.0, 5.0]))
    max_exp, L, T = int(r.choice([10, 6, 2])), int(r.choice([0, 1, 7, 20])), int(r.choice([1, 3, 6]))
    # per-chain step sizes / diagonal metrics (what window adaptation hands back), and sample() instead of calls
    per_chain = bool((mk == "diag" and r.random() < 0.3) or (mk == "dense" and D <= 130 and r.random() < 0.25))
    use_sample = bool(T > 1 and r.random() < 0.5)
    desc = dict(case=case, sampler=sampler, tk=tk, mk=mk, D=D, C=C, eps=eps, thr=thr, max_exp=max_exp, L=L, T=T,
                per_chain=per_chain, use_sample=use_sample, **opts)
    if tk == "linreg" and mk == "scalar":
        return None
    seeds = [int(x) for x in r.integers(0, 2 ** 31, size=C)]
    for k, v in opts.items():
        eng.set_option(k, v)
    try:
        q, U, g = co.new_state(otgt, q0.copy())
        srng = RandomStream(seeds=seeds)
        dq0 = torch.as_tensor(q0, device="cuda")
        if per_chain:
            eps_c = eps * (0.5 + r.random(C))
            if mk == "dense" and tk == "linreg":  # the shared 2 x 2 matrix, rescaled and tilted per chain
                sc = 0.5 + r.random((C, 1, 1))
                tilt = 0.3 * (r.random(C) - 0.5)
                imm_c = np.asarray(imm)[None] * sc
                imm_c[:, 0, 1] += tilt * np.sqrt(imm_c[:, 0, 0] * imm_c[:, 1, 1])
                imm_c[:, 1, 0] = imm_c[:, 0, 1]
            elif mk == "dense":  # every chain its own dense matrix (what is_mass_matrix_full adaptation hands back)
                Bc = r.normal(size=(C, D, D))
                imm_c = Bc @ Bc.transpose(0, 2, 1) / D + 0.5 * np.eye(D)
                imm_c = 0.5 * (imm_c + imm_c.transpose(0, 2, 1))
            else:
                imm_c = np.asarray(imm)[None, :] * (0.5 + r.random((C, D)))
            g_eps, g_imm = PerChain(torch.as_tensor(eps_c, device="cuda")), PerChain(torch.as_tensor(imm_c, device="cuda"))
        else:
            eps_c, imm_c, g_eps, g_imm = None, None, eps, imm
        nuts_ = sampler == "nuts"
        rng = co.site_states(seeds, 4 if nuts_ else 2)

        def oracle_step():
            if not per_chain:
                m_ = co.Metric(imm, D)
                return (co.nuts_step(otgt, m_, rng, eps, q, U, g, max_exp=max_exp, thr=thr) if nuts_
                        else co.hmc_step(otgt, m_, rng, eps, L, q, U, g, thr=thr))
            outs = []
            for c in range(C):  # the oracle takes one step size / metric per call
                qc, Uc, gc, rc = q[c:c + 1].copy(), U[c:c + 1].copy(), g[c:c + 1].copy(), rng[c:c + 1].copy()
                m_ = co.Metric(imm_c[c], D)
                o = (co.nuts_step(otgt, m_, rc, float(eps_c[c]), qc, Uc, gc, max_exp=max_exp, thr=thr) if nuts_
                     else co.hmc_step(otgt, m_, rc, float(eps_c[c]), L, qc, Uc, gc, thr=thr))
                q[c], U[c], g[c], rng[c] = qc[0], Uc[0], gc[0], rc[0]
                outs.append(o)
            return {k: np.concatenate([np.atleast_1d(o[k]) for o in outs]) for k in outs[0] if k != "momentum"}

        kernel = (nuts.new_kernel(srng, tgt, max_num_expansions=max_exp, divergence_threshold=thr) if nuts_
                  else hmc.new_kernel(srng, tgt, divergence_threshold=thr))
        state = (nuts if nuts_ else hmc).new_state(dq0, tgt)
        extra = () if nuts_ else (L,)
        if use_sample:
            samples, info, acc, div = kernel.sample(state, g_eps, g_imm, *extra, T)
            for t in range(T):
                res = oracle_step()
                np.testing.assert_allclose(samples[t].cpu().numpy().reshape(q.shape), q, rtol=RTOL, atol=1e-10)
                assert np.array_equal(div[t].cpu().numpy().reshape(-1).astype(np.int64), res["is_diverging"].astype(np.int64))
            holder = (kernel._nuts if nuts_ else kernel._hmc)["holder"]["rng"]
        else:
            for _ in range(T):
                info, upd = kernel(state, g_eps, g_imm, *extra)
                state = info.state._replace(momentum=None)
                res = oracle_step()
                for f in (("n_leapfrog", "num_doublings", "is_turning", "is_diverging") if nuts_ else ("is_diverging",)):
                    got = getattr(info, f).cpu().numpy().reshape(-1).astype(np.int64)
                    assert np.array_equal(got, res[f].astype(np.int64)), (f, got.tolist(), res[f].tolist())
            holder = upd[srng]
        pos = info.state.position.cpu().numpy().reshape(q.shape)
        np.testing.assert_allclose(pos, q, rtol=RTOL, atol=1e-10)
        np.testing.assert_allclose(info.state.potential_energy.cpu().numpy().reshape(-1), U, rtol=RTOL, atol=1e-10)
        # (exp of an energy DIFFERENCE: rounding of energies of size 1e3-1e4 is amplified -- the north star's 1e-6 here)
        np.testing.assert_allclose(info.acceptance_probability.cpu().numpy().reshape(-1), res["acceptance_probability"], rtol=1e-6, atol=1e-12)
        got = holder.cpu().numpy().view(np.uint64).reshape(rng.shape)
        assert np.array_equal(got[:, :, :2], rng[:, :, :2]), "generator state"
    finally:
        for k, v in (("resident_nuts", 2), ("resident_min_team", 0), ("fused_hmc", 1), ("dense_linear", 1), ("fused_nuts", 0),
                     ("streamk", 2), ("compact", 1), ("block_dense", 1), ("block_roll", 0)):
            eng.set_option(k, v)
    return desc


t0, n, bad = time.time(), 0, []


def cases():
    """FUZZ_CASES=id,id,... : exactly those; FUZZ_COUNT=N : the N ids from seed * 10**6 on (both deterministic -- what the test
    suite runs); otherwise ids from seed * 10**6 on for `seconds` of wall clock (the long sweeps recorded under profiles/)."""
    only = [int(x) for x in os.environ.get("FUZZ_CASES", "").split(",") if x]
    if only:
        yield from only
    elif os.environ.get("FUZZ_COUNT"):
        yield from range(seed0 * 1_000_000, seed0 * 1_000_000 + int(os.environ["FUZZ_COUNT"]))
    else:
        case = seed0 * 1_000_000
        while time.time() - t0 < budget:
            yield case
            case += 1


for case in cases():
    try:
        d = one(case)
        n += d is not None
    except Exception as e:
        bad.append((case, repr(e)[:400]))
        print("MISMATCH case", case, repr(e)[:700], flush=True)
        traceback.print_exc(limit=1)
print(f"fuzz: {n} configurations in {time.time() - t0:.0f} s, {len(bad)} mismatches: {[b[0] for b in bad]}")
sys.exit(1 if bad else 0)
