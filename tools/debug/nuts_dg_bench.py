#!/usr/bin/env python3
"""Diagonal-Gaussian target (separate dU/dq, three parameter vectors) on the workgroup-per-chain NUTS kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
D = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = np.random.default_rng(0)
mu, sigma = r.normal(size=D), 0.5 + r.random(D)
imm = torch.as_tensor(sigma ** 2, device="cuda")
q0 = torch.as_tensor(mu + sigma * r.standard_normal((C, D)), device="cuda")
tgt = targets.DiagGaussian(mu, sigma)
eps = 0.5 * D ** -0.25
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = nuts.new_state(q0, tgt)
info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None)
torch.cuda.synchronize(); t0 = time.perf_counter(); nl = 0
for _ in range(3):
    info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None)
    nl += int(info.n_leapfrog.sum().item())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"DiagGaussian D={D} C={C}: {nl/dt:.3e} leapfrog/s {dt/3*1e3:.2f} ms/transition {nl/3/C:.1f} leapfrogs/chain")
