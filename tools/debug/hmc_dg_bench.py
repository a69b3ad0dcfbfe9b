#!/usr/bin/env python3
"""Diagonal-Gaussian target on the workgroup-per-chain HMC kernel (separate dU/dq, parameters streamed from L2)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets
D = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = np.random.default_rng(0)
mu, sigma = r.normal(size=D), 0.5 + r.random(D)
imm = torch.as_tensor(sigma ** 2, device="cuda")
q0 = torch.as_tensor(mu + sigma * r.standard_normal((C, D)), device="cuda")
tgt = targets.DiagGaussian(mu, sigma)
eps = 0.5 * D ** -0.25
kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = hmc.new_state(q0, tgt)
_, info, _, _ = kernel.sample(state, eps, imm, 32, 2, keep_samples=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, info, acc, _ = kernel.sample(info.state._replace(momentum=None), eps, imm, 32, T, keep_samples=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"DiagGaussian HMC D={D} C={C}: {C*32*T/dt:.3e} leapfrog/s {dt/T*1e3:.2f} ms/transition accept {acc.mean().item():.3f}")
