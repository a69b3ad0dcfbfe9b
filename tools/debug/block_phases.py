#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_nuts_block_dense (needs `make -C aehmc_amd/csrc timing`; run with
AEHMC_AMD_LIB=aehmc_amd/libaehmc_hip_timing.so).  usage: block_phases.py [D] [C] [E]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
E = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = np.random.default_rng(0)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
eng = get_engine()
if os.environ.get("BLOCK_DENSE"):
    eng.set_option("block_dense", int(os.environ["BLOCK_DENSE"]))
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=E)
state = nuts.new_state(q0, tgt)
eps = 0.3 * D ** -0.25
for _ in range(2):
    info, _ = kernel(state, eps, imm)
    state = info.state._replace(momentum=None)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
info, _ = kernel(state, eps, imm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ws = eng._ws
vec = ((C * D * 8) + 255) & ~255
off = (26 + 3 * E) * vec
tim = ws[off: off + C * 8 * 8].view(torch.float64).reshape(C, 8).cpu().numpy()
nl = info.n_leapfrog.cpu().numpy()
blocks = nl[: C // 16 * 16].reshape(-1, 16).max(axis=1)
names = ["rows->LDS", "barrier A", "MFMA", "barrier B", "book", "leap12", "vote", "begin"]
if D <= 256 and os.environ.get("BLOCK_DENSE", "1") == "1":  # the register kernel's phases
    names = ["book: pass", "book: scal", "MFMA", "barriers", "book: rest", "stage12", "vote", "begin"]
if os.environ.get("BLOCK_DENSE"):
    pass
tot = tim.sum(axis=1)
print(f"D={D} C={C}: one transition {dt*1e3:.3f} ms; leapfrogs/chain mean {nl.mean():.1f} max {nl.max()}; "
      f"per workgroup max: mean {blocks.mean():.1f}; cycles per wave total {tot.mean():.0f}")
steps = np.repeat(blocks, 16)[: len(tot)]
for k, n in enumerate(names):
    print(f"  {n:10s} {np.mean(tim[:len(steps), k] / steps):9.0f} ticks / workgroup step  {100 * tim[:, k].sum() / tot.sum():5.1f} %")
