#!/usr/bin/env python3
"""NUTS with a dense metric at small D (the classic full-mass-matrix use): wall time per lock-step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = nuts.new_state(torch.as_tensor(r.standard_normal((C, D)), device="cuda"), tgt)
for _ in range(3):
    info, _ = kernel(state, 0.3 * D ** -0.25, imm); state = info.state._replace(momentum=None)
torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 0
for _ in range(10):
    info, _ = kernel(state, 0.3 * D ** -0.25, imm); state = info.state._replace(momentum=None)
    steps += int(info.n_leapfrog.max().item())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"D={D} C={C}: {dt/10*1e3:.3f} ms/transition, {steps/10:.1f} lock-steps (deepest tree), {dt/steps*1e6:.1f} us per lock-step")
