#!/usr/bin/env python3
"""Time of the notebook's HMC transition (G2: 1 chain, 1e4 rows, L = 1024) on the fused and lock-step paths."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets
from aehmc_amd.engine import get_engine
eng = get_engine()
rng = np.random.default_rng(0)
X = rng.normal(0, 1, size=(10_000,)); y = 3 * X + rng.normal(0, 1)
tgt = targets.LinearRegression(X, y)
for C in (1, 1024):
    for fused in (1, 0):
        eng.set_option("fused_hmc", fused)
        q0 = np.tile(np.array([3.0, np.log(0.21)]), (C, 1))
        kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        state = hmc.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        info, _ = kernel(state, 5e-5, np.array([1.0, 1.0]), 1024)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        info, _ = kernel(state, 5e-5, np.array([1.0, 1.0]), 1024)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"C={C} fused={fused}: {dt*1e3:.2f} ms per transition (L=1024), {C*1024/dt:.3e} leapfrog/s, q[0]={info.state.position[0].cpu().numpy()}")
eng.set_option("fused_hmc", 1)
