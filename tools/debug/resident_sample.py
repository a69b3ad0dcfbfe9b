#!/usr/bin/env python3
"""kernel.sample(T) of diagonal-metric NUTS on the register-resident kernels: ms per transition and leapfrog/s.
usage: resident_sample.py [DxC ...] (default 50x4096 100x4096 200x4096); T = 100 transitions per launch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets

T = int(os.environ.get("T", 100))
r = np.random.default_rng(0)
cases = [tuple(int(v) for v in x.split("x")) for x in sys.argv[1:]] or [(50, 4096), (100, 4096), (200, 4096)]
for D, C in cases:
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    tgt = targets.DiagGaussian(mu, sigma)
    q0 = torch.as_tensor(mu + sigma * r.standard_normal((C, D)), device="cuda")
    imm = torch.as_tensor(sigma ** 2, device="cuda")
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    out = kernel.sample(state, 0.5 * D ** -0.25, imm, 5)
    best, nl = 1e9, 0
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = kernel.sample(out[1].state._replace(momentum=None), 0.5 * D ** -0.25, imm, T)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if dt < best: best, nl = dt, int(out[1].n_leapfrog.sum().item())
    print(f"DiagGaussian D={D} C={C} sample({T}): {best / T * 1e3:8.4f} ms/transition {nl / best:10.3e} leapfrog/s", flush=True)
