#!/usr/bin/env python3
"""resident_nuts 1 (always) vs 0 (lock-step) on small-D NUTS with a few thousand chains."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
eng = get_engine()
r = np.random.default_rng(0)
def run(name, tgt, q0, eps, imm, n=5):
    C = q0.shape[0]
    for opt in (1, 0, 2):
        eng.set_option("resident_nuts", opt)
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None)
        torch.cuda.synchronize(); t0 = time.perf_counter(); nl = torch.zeros((), dtype=torch.int64, device="cuda"); mx = 0
        for _ in range(n):
            info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None); nl += info.n_leapfrog.sum()
            mx = max(mx, int(info.n_leapfrog.max().item()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:44s} resident_nuts={opt}: {int(nl.item())/dt:10.3e} leapfrog/s {dt/n*1e3:8.3f} ms/transition, deepest tree {mx}", flush=True)
    eng.set_option("resident_nuts", 2)
import itertools
cases = [(int(a), int(b)) for a, b in (x.split("x") for x in sys.argv[1:])] if len(sys.argv) > 1 else \
    [(10, 4096), (10, 8192), (50, 4096), (100, 4096), (100, 8192), (256, 4096), (256, 12000)]
run("warm-up (first use of every kernel)", targets.IsoGaussian(), r.standard_normal((3000, 7)), 0.3, np.ones(7), n=2)
for D, C in cases:
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    q0 = mu + sigma * r.standard_normal((C, D))
    run(f"DiagGaussian D={D} C={C}", targets.DiagGaussian(mu, sigma), q0, 0.5 * D ** -0.25, sigma ** 2)
for D, C in ((100, 4096), (256, 8192)):
    run(f"IsoGaussian D={D} C={C}", targets.IsoGaussian(), r.standard_normal((C, D)), 0.5 * D ** -0.25, np.ones(D))
