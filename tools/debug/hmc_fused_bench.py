#!/usr/bin/env python3
"""Diagonal-metric HMC on the register-resident kernel (k_hmc_fused): usage hmc_fused_bench.py D [iso|diag] [fp_contract] [C] [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "iso"; fc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096; T = int(sys.argv[5]) if len(sys.argv) > 5 else 100
get_engine().set_option("fp_contract", fc)
r = np.random.default_rng(0)
mu, sigma = r.normal(size=D), 0.5 + r.random(D)
tgt = targets.IsoGaussian() if kind == "iso" else targets.DiagGaussian(mu, sigma)
imm = torch.ones(D, dtype=torch.float64, device="cuda") if kind == "iso" else torch.as_tensor(sigma ** 2, device="cuda")
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = hmc.new_state(q0, tgt)
_, info, _, _ = kernel.sample(state, 0.1, imm, 32, 5, keep_samples=False)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, info, acc, _ = kernel.sample(info.state._replace(momentum=None), 0.1, imm, 32, T, keep_samples=False)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"HMC L=32 D={D} {kind} fp_contract={fc} C={C}: {C*32*T/best:.3e} leapfrog/s {best/T*1e3:.3f} ms/transition accept {acc.mean().item():.3f}", flush=True)
