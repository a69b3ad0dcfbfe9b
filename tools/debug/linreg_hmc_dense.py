#!/usr/bin/env python3
"""HMC on the regression target with a dense 2 x 2 metric: one-launch kernel (fused_hmc = 1) against lock-step (0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets
from aehmc_amd.engine import get_engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
r = np.random.default_rng(0)
X = r.normal(size=N); y = 3 * X + 0.5 * r.normal(size=N)
tgt = targets.LinearRegression(X, y)
imm = np.array([[1.0 / N, 0.2 / N], [0.2 / N, 0.5 / N]])
q0 = torch.as_tensor(np.array([3.0, np.log(0.5)]) + 0.02 * r.normal(size=(C, 2)), device="cuda")
eng = get_engine()
for mode in (1, 0):
    eng.set_option("fused_hmc", mode)
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(q0, tgt)
    kernel.sample(state, 0.5, imm, L, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info, acc, _ = kernel.sample(state, 0.5, imm, L, 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fused_hmc={mode} N={N} C={C} L={L}: {dt/10*1e3:.3f} ms/transition, acceptance {float(acc.mean()):.2f}", flush=True)
eng.set_option("fused_hmc", 1)
