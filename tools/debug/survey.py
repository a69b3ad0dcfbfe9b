#!/usr/bin/env python3
"""Throughput survey over user-like configurations (not the benchmarked ones): spots pathologies."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets

def run(name, mod, tgt, q0, eps, imm, extra=(), n=5):
    C = q0.shape[0]
    kernel = mod.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = mod.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    info, _ = kernel(state, eps, imm, *extra); state = info.state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nl = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(n):
        info, _ = kernel(state, eps, imm, *extra); state = info.state._replace(momentum=None); nl += info.n_leapfrog.sum()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:58s} {int(nl.item())/dt:10.3e} leapfrog/s {dt/n*1e3:9.3f} ms/transition  {int(nl.item())/n/C:6.1f} leapfrogs/chain", flush=True)

r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
for D, C in ((10, 4096), (50, 4096), (50, 65536), (200, 4096), (1000, 4096), (3000, 1024)):
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    q0 = mu + sigma * r.standard_normal((C, D))
    run(f"NUTS DiagGaussian diag-metric D={D} C={C}", nuts, targets.DiagGaussian(mu, sigma), q0, 0.5 * D ** -0.25, sigma ** 2)
for D, C in ((10, 4096), (50, 4096), (200, 4096), (1000, 1024), (3000, 512)):
    P, imm = spd(D), spd(D)
    q0 = r.standard_normal((C, D))
    run(f"NUTS DenseMVN dense-metric D={D} C={C}", nuts, targets.DenseMVN(np.zeros(D), P), q0, 0.3 * D ** -0.25, imm)
    run(f"HMC L=16 DenseMVN dense-metric D={D} C={C}", hmc, targets.DenseMVN(np.zeros(D), P), q0, 0.3 * D ** -0.25, imm, (16,))
for D, C in ((50, 4096), (500, 4096)):
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    q0 = mu + sigma * r.standard_normal((C, D))
    run(f"NUTS DiagGaussian dense-metric D={D} C={C}", nuts, targets.DiagGaussian(mu, sigma), q0, 0.3 * D ** -0.25, spd(D))
    run(f"HMC L=16 DiagGaussian diag-metric D={D} C={C}", hmc, targets.DiagGaussian(mu, sigma), q0, 0.3 * D ** -0.25, sigma ** 2, (16,))
