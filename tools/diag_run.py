#!/usr/bin/env python3
"""Small fixed workload for rocprofv3: diagonal-mass NUTS and HMC at D = 1e4, 4096 chains.
usage: python3 tools/diag_run.py [nuts|hmc|both] [transitions] [D] [C]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, hmc, nuts, targets

what = sys.argv[1] if len(sys.argv) > 1 else "both"
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
eps = 0.5 * D ** -0.25
q0 = torch.as_tensor(np.random.default_rng(0).standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
tgt = targets.IsoGaussian()
if what in ("nuts", "both"):
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    nl = 0
    for _ in range(NT):
        info, _ = kernel(state, eps, imm)
        state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum().item())
    print(f"nuts D={D} C={C}: {nl} leapfrogs in {NT} transitions")
if what in ("hmc", "both"):
    kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = hmc.new_state(q0, tgt)
    _, info, acc, _ = kernel.sample(state, eps, imm, 32, NT, keep_samples=False)
    torch.cuda.synchronize()
    print(f"hmc D={D} C={C}: {C * 32 * NT} leapfrogs in {NT} transitions, accept {acc.mean().item():.3f}")
