#!/usr/bin/env python3
"""Regression target (N rows) with is_mass_matrix_full warm-up (per-chain dense 2 x 2 metric: lock-step path)
against the diagonal warm-up (one launch of k_nuts_linreg)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
r = np.random.default_rng(0)
X = r.normal(size=N); y = 3 * X + r.normal()
tgt = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * r.normal(size=(C, 2))
for full in (False, True):
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, (eps, imm), _ = window_adaptation.run(kernel, state, W, is_mass_matrix_full=full)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    samples, info = kernel.sample(state, eps, imm, 50)[:2]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"full={full} N={N} C={C}: {W} warm-up steps {t1-t0:.3f} s, 50 samples {t2-t1:.3f} s, "
          f"{float(info.n_leapfrog.double().mean())/50:.1f} leapfrogs/transition", flush=True)
