#!/usr/bin/env python3
"""window_adaptation.run around an HMC kernel: one C-ABI call (fused) against the caller-side Python loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets, window_adaptation
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
r = np.random.default_rng(0)
mu, sigma = r.normal(size=D), 0.5 + r.random(D)
tgt = targets.DiagGaussian(mu, sigma)
q0 = torch.as_tensor(mu + sigma * r.standard_normal((C, D)), device="cuda")
for fused in (True, False, True, False):
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(q0, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, (eps, imm), _ = window_adaptation.run(kernel, state, W, num_integration_steps=L, fused=fused)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fused={fused} D={D} C={C} L={L}: {W} warm-up steps {dt:.3f} s ({dt/W*1e6:.0f} us per step), median eps {float(eps.value.median()):.3f}", flush=True)
