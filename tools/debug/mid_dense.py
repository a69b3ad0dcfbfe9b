#!/usr/bin/env python3
"""NUTS / HMC with a shared dense metric and a dense-precision target at MID-size D (65 .. ~1000: lock-step path,
GEMMs of C x D x D): time per transition of kernel.sample(N)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
HMC_L = int(sys.argv[4]) if len(sys.argv) > 4 else 0
r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
mod, extra = (hmc, (HMC_L,)) if HMC_L else (nuts, ())
if os.environ.get("BLOCK_DENSE"):
    from aehmc_amd.engine import get_engine
    get_engine().set_option("block_dense", int(os.environ["BLOCK_DENSE"]))
if os.environ.get("BLOCK_ROLL"):
    from aehmc_amd.engine import get_engine
    get_engine().set_option("block_roll", int(os.environ["BLOCK_ROLL"]))
kernel = mod.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = mod.new_state(q0, tgt)
samples, info = kernel.sample(state, 0.3 * D ** -0.25, imm, *extra, 3)[:2]
state = info.state._replace(momentum=None)
dt = 1e9
for _ in range(3):  # (the first call at a new N also allocates its [N, C, D] sample buffer)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info = kernel.sample(state, 0.3 * D ** -0.25, imm, *extra, N)[:2]
    torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    state = info.state._replace(momentum=None)
nl = float(HMC_L) if HMC_L else float(info.n_leapfrog.double().mean()) / N
name = "hmc" if HMC_L else "nuts"
print(f"{name} D={D} C={C}: {dt/N*1e3:.3f} ms/transition, {nl:.1f} leapfrogs/transition/chain, {C*nl*N/dt:.3e} leapfrog/s", flush=True)
