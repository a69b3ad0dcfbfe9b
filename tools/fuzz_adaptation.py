#!/usr/bin/env python3
"""Randomised check of window_adaptation.run: the warm-up in one C-ABI call (one launch where the kernel
family allows it) against the step-by-step loop (one transition + one adaptation update per step):
identical state, step sizes, (inverse) mass matrices and generator state.  usage: fuzz_adaptation.py [seconds] [seed]"""
import os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine
from oracle import c_oracle as co, np_adaptation as na, np_oracle as no

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine()


class OracleNuts:
    """kernel(state, step_size, imm) for one chain, backed by oracle/c (scheme-A RNG)."""

    def __init__(self, otgt, seed, D, max_exp):
        self.otgt, self.D, self.max_exp = otgt, D, max_exp
        self.rng = co.site_states([seed], 4)

    def __call__(self, state, eps, imm):
        q = np.asarray(state.position, dtype=np.float64).reshape(1, self.D).copy()
        U = np.array([state.potential_energy], dtype=np.float64)
        g = np.asarray(state.potential_energy_grad, dtype=np.float64).reshape(1, self.D).copy()
        res = co.nuts_step(self.otgt, co.Metric(imm, self.D), self.rng, float(eps), q, U, g, max_exp=self.max_exp)
        return SimpleNamespace(state=no.IntegratorState(q[0].copy(), None, float(U[0]), g[0].copy()),
                               acceptance_probability=float(res["acceptance_probability"][0]))


def one(case):
    r = np.random.default_rng(case)
    kind = r.choice(["diag", "linreg", "std"], p=[0.6, 0.25, 0.15])
    full = bool(r.random() < 0.25)
    steps = int(r.choice([1, 5, 21, 37, 75, 130, 160]))
    C = int(r.choice([1, 2, 5, 9, 33, 70]))
    opts = {"resident_nuts": int(r.choice([0, 1, 2])), "resident_min_team": int(r.integers(0, 2))}
    if kind == "linreg":
        N = int(r.choice([500, 10000, 12001]))
        X = r.normal(size=N); y = 3 * X + 0.5 * r.normal(size=N)
        tgt, D = targets.LinearRegression(X, y), 2
        q0 = np.array([3.0, np.log(0.5)]) + 0.02 * r.normal(size=(C, 2))
    else:
        D = int(r.choice([1, 2, 3, 7, 16, 24, 40, 64, 65, 100, 200, 400, 700] if not full else [2, 3, 7, 16, 40, 70]))
        mu, sigma = r.normal(size=D), 0.5 + 2 * r.random(D)
        tgt = targets.DiagGaussian(mu, sigma) if kind == "diag" else targets.StdNormal()
        q0 = (mu + sigma * r.normal(size=(C, D))) if kind == "diag" else r.normal(size=(C, D))
        if D > 400:
            C = min(C, 5)
            q0 = q0[:C]
    for k, v in opts.items():
        eng.set_option(k, v)
    outs = []
    try:
        for fused in (True, False):
            srng = RandomStream(seeds=[int(x) for x in np.random.default_rng(case + 1).integers(0, 2 ** 31, size=C)])
            kernel = nuts.new_kernel(srng, tgt, max_num_expansions=int(np.random.default_rng(case + 2).choice([10, 5])))
            state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
            state, (eps, imm), upd = window_adaptation.run(kernel, state, steps, is_mass_matrix_full=full, fused=fused)
            info, upd = kernel(state, eps, imm)
            outs.append((state.position.clone(), state.potential_energy.clone(), eps.value.clone(), imm.value.clone(),
                         imm.sqrt_mass.clone(), info.state.position.clone(), info.n_leapfrog.clone(), upd[srng].clone()))
    finally:
        eng.set_option("resident_nuts", 2)
        eng.set_option("resident_min_team", 0)
    for k, (a, b) in enumerate(zip(*outs)):
        same = torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        assert same, (k, dict(case=case, kind=kind, full=full, steps=steps, C=C, D=D, **opts))
    # short warm-ups of small diagonal problems also against the restatements (numpy adaptation + C oracle NUTS);
    # the loop feeds the step size back into the trajectory, so only short horizons are comparable
    if kind == "diag" and not full and D <= 7 and steps <= 37 and C <= 9:  # (at 75 steps single chains drift to 1e-5)
        max_exp = int(np.random.default_rng(case + 2).choice([10, 5]))
        seeds = [int(x) for x in np.random.default_rng(case + 1).integers(0, 2 ** 31, size=C)]
        otgt = co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
        eps_g, imm_g, pos_g = outs[0][2].cpu().numpy().reshape(-1), outs[0][3].cpu().numpy().reshape(C, D), outs[0][0].cpu().numpy().reshape(C, D)
        for c in range(C):
            Uo, go = no.DiagGaussian(mu, sigma)(q0[c])
            st, (eps_o, imm_o) = na.run(OracleNuts(otgt, seeds[c], D, max_exp), no.IntegratorState(q0[c], None, Uo, go), steps)
            err = max(abs(eps_g[c] / eps_o - 1), np.abs(imm_g[c] / np.asarray(imm_o).reshape(-1) - 1).max())
            if os.environ.get("FUZZ_VERBOSE"):
                print("   oracle cmp", dict(case=case, steps=steps, D=D, C=C, c=c, max_exp=max_exp, **opts), "rel err", err)
            # (typical 1e-11 .. 1e-13; the loop feeds the step size back into the trajectory, so a rounding difference in one
            #  sum is amplified along the warm-up: single chains reach 2e-6 .. 3.5e-5 at 37 steps -- one in 4.4e3
            #  configurations in round 3, its sibling chain at 1e-13 --, 8e-5 at 75)
            np.testing.assert_allclose(eps_g[c], eps_o, rtol=1e-4)
            np.testing.assert_allclose(imm_g[c], np.asarray(imm_o).reshape(-1), rtol=1e-4)
            np.testing.assert_allclose(pos_g[c], st.position, rtol=1e-4, atol=1e-4 * np.abs(st.position).max())  # (a coordinate near 0)
        global n_oracle
        n_oracle += 1


n_oracle = 0
t0, n, bad = time.time(), 0, []
case = seed0 * 1_000_000
only = [int(x) for x in os.environ.get("FUZZ_CASES", "").split(",") if x]
for case in only:
    try:
        one(case)
    except Exception as e:
        print("MISMATCH case", case, repr(e)[:300], flush=True)
if only:
    sys.exit(0)
while time.time() - t0 < budget:
    try:
        one(case)
        n += 1
    except Exception as e:
        bad.append(case)
        print("MISMATCH case", case, repr(e)[:600], flush=True)
        traceback.print_exc(limit=1)
    case += 1
print(f"fuzz_adaptation: {n} configurations in {time.time() - t0:.0f} s ({n_oracle} of them also against the restatements), {len(bad)} mismatches: {bad}")
sys.exit(1 if bad else 0)
