#!/usr/bin/env python3
"""NUTS with a shared dense metric and a dense-precision target at small D (the classic full-mass-matrix use):
time per transition of kernel.sample(100), single-launch kernel (resident_nuts = 2) against lock-step (0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
MODES = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 0]
HMC_L = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # > 0: HMC with that trajectory length instead of NUTS
r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
eng = get_engine()
for mode in MODES:
    eng.set_option("resident_nuts", mode)
    eng.set_option("fused_hmc", 1 if mode else 0)
    mod, extra = (hmc, (HMC_L,)) if HMC_L else (nuts, ())
    kernel = mod.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = mod.new_state(q0, tgt)
    samples, info = kernel.sample(state, 0.3 * D ** -0.25, imm, *extra, 20)[:2]
    state = info.state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info = kernel.sample(state, 0.3 * D ** -0.25, imm, *extra, 100)[:2]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nl = float(HMC_L) if HMC_L else float(info.n_leapfrog.double().mean()) / 100
    name = "hmc" if HMC_L else "nuts"
    print(f"{name} single_launch={int(bool(mode))} D={D} C={C}: {dt/100*1e3:.3f} ms/transition, {nl:.1f} leapfrogs/transition/chain, "
          f"{C*nl*100/dt:.3e} leapfrog/s", flush=True)
