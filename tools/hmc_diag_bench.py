#!/usr/bin/env python3
"""Throughput of diagonal-metric HMC at large D (resident kernel vs lock-step streaming).
usage: python tools/hmc_diag_bench.py D C [L] [transitions]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, hmc, targets
from aehmc_amd.engine import get_engine

D, C = int(sys.argv[1]), int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 10
eng = get_engine()
q0 = torch.as_tensor(np.random.default_rng(0).standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
eps = 0.5 * D ** -0.25
for fused in (1, 0):
    eng.set_option("fused_hmc", fused)
    tgt = targets.IsoGaussian()
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(q0, tgt)
    _, info, _, _ = kernel.sample(state, eps, imm, L, 2, keep_samples=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, info, acc, _ = kernel.sample(info.state._replace(momentum=None), eps, imm, L, NT, keep_samples=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nl = C * L * NT
    print(f"D={D} C={C} L={L} resident/fused={fused}: {nl/dt:.3e} leapfrog/s  {dt/NT*1e3:.2f} ms/transition  "
          f"accept {acc.mean().item():.3f}  ~{48.0*D*nl/dt/1e9:.0f} GB/s at 48*D B/leapfrog")
eng.set_option("fused_hmc", 1)
