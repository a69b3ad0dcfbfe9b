#!/usr/bin/env python3
"""Randomised check of window_adaptation.run: the warm-up in one C-ABI call (one launch where the kernel
family allows it) against the step-by-step loop (one transition + one adaptation update per step):
identical state, step sizes, (inverse) mass matrices and generator state.  usage: fuzz_adaptation.py [seconds] [seed]"""
import os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine()


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


t0, n, bad = time.time(), 0, []
case = seed0 * 1_000_000
while time.time() - t0 < budget:
    try:
        one(case)
        n += 1
    except Exception as e:
        bad.append(case)
        print("MISMATCH case", case, repr(e)[:600], flush=True)
        traceback.print_exc(limit=1)
    case += 1
print(f"fuzz_adaptation: {n} configurations in {time.time() - t0:.0f} s, {len(bad)} mismatches: {bad}")
sys.exit(1 if bad else 0)
