"""Warm-up wall time: window_adaptation.run in one launch (fused) vs the step-by-step loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
for C, D, steps in [(64, 10, 1000), (1024, 100, 1000), (4096, 100, 300), (16384, 16, 300)]:
    r = np.random.default_rng(D)
    mu, sigma = r.normal(size=D), 0.5 + 2 * r.random(D)
    tgt = targets.DiagGaussian(mu, sigma)
    q0 = mu + sigma * r.normal(size=(C, D))
    for fused in (True, False):
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, (eps, imm), _ = window_adaptation.run(kernel, state, steps, fused=fused)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"C={C} D={D} {steps} warm-up steps fused={fused}: {dt*1e3:.1f} ms; median step size {eps.value.median().item():.3f}")
