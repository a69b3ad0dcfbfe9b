#!/usr/bin/env python3
"""Throughput of diagonal-metric NUTS: fused single-launch kernel vs lock-step path.
usage: python tools/nuts_diag_bench.py D C [eps] [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine

D, C = int(sys.argv[1]), int(sys.argv[2])
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5 * D ** -0.25
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
eng = get_engine()
q0 = torch.as_tensor(np.random.default_rng(0).standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
for fused in (1, 0):
    eng.set_option("resident_nuts", fused)
    tgt = targets.IsoGaussian()
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nl = 0
    for _ in range(steps):
        info, _ = kernel(state, eps, imm); state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum().item())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"D={D} C={C} eps={eps:.4f} resident={fused}: {nl/dt:.3e} leapfrog/s  {dt/steps*1e3:.2f} ms/transition "
          f"{nl/steps/C:.1f} leapfrogs/chain  ~{88.0*D*nl/dt/1e9:.0f} GB/s at 88*D B/leapfrog")
