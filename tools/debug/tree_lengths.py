#!/usr/bin/env python3
"""Leapfrogs per chain and transition of the mid-size dense workload (tools/debug/mid_dense.py) -> gpurun_out/nleap_D.npy
[transitions, chains]: input of the roll-on scheduling estimate in profiles/r4/INDEX.md."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 60
r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = nuts.new_state(q0, tgt)
out = []
for t in range(N + 3):
    info, _ = kernel(state, 0.3 * D ** -0.25, imm)
    state = info.state._replace(momentum=None)
    if t >= 3: out.append(info.n_leapfrog.cpu().numpy().astype(np.int32))
out = np.stack(out)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/nleap_{D}.npy", out)
print(D, out.shape, out.mean(), out.max(), np.bincount(out.ravel())[:70])
