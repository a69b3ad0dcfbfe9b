#!/usr/bin/env python3
"""NUTS with one dense inverse mass matrix per chain (what full-matrix window adaptation returns), coordinate-wise
target: ms per transition, leapfrog/s and the HBM rate on the matrix bytes (one product w' = imm dU/dq' per leapfrog in
linear dense mode + three at the start of a transition: D^2 x 8 bytes each).  usage: pc_dense_time.py [D] [C] [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import PerChain, RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = get_engine()
if os.environ.get("PC_DENSE"):
    eng.set_option("pc_dense", int(os.environ["PC_DENSE"]))
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(C, D, D, dtype=torch.float64, device="cuda", generator=g)
imm = torch.baddbmm(0.3 * torch.eye(D, dtype=torch.float64, device="cuda").expand(C, D, D), A, A.transpose(1, 2), alpha=1.0 / D)
imm = 0.5 * (imm + imm.transpose(1, 2))
del A
r = np.random.default_rng(0)
tgt = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = nuts.new_state(q0, tgt)
pi = PerChain(imm)  # (a device tensor: factored once, the factors are kept while it is unchanged)
eps = 0.25 * D ** -0.25
samples, info = kernel.sample(state, eps, pi, 2)[:2]
state = info.state._replace(momentum=None)
dt = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info = kernel.sample(state, eps, pi, T)[:2]
    torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    state = info.state._replace(momentum=None)
nl = float(info.n_leapfrog.double().sum())
bytes_ = (nl + 3.0 * C * T) * D * D * 8
print(f"per-chain dense NUTS D={D} C={C}: {dt / T * 1e3:.3f} ms/transition, {nl / C / T:.1f} leapfrogs/transition/chain, "
      f"{nl / dt:.3e} leapfrog/s, {bytes_ / dt / 1e12:.2f} TB/s on the matrix bytes", flush=True)
