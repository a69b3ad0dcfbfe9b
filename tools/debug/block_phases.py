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
T = int(os.environ.get("T", 1))
if os.environ.get("BLOCK_ROLL"):
    eng.set_option("block_roll", int(os.environ["BLOCK_ROLL"]))
import time
if T > 1:
    out = kernel.sample(state, eps, imm, 3)
    state = out[1].state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = kernel.sample(state, eps, imm, T)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / T
    info = out[1]
    nl = info.n_leapfrog.cpu().numpy() / T
else:
    for _ in range(2):
        info, _ = kernel(state, eps, imm)
        state = info.state._replace(momentum=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, _ = kernel(state, eps, imm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nl = info.n_leapfrog.cpu().numpy()
ws = eng._ws
vec = ((C * D * 8) + 255) & ~255
off = (26 + 3 * E) * vec
flow = False  # (k_nuts_block_flow: tools/debug/experiments/block_flow)
nrec = (C + 15) // 16 * 4 if flow else C  # k_nuts_block_flow: one record per wavefront (4 per workgroup)
NS = 16 if (D <= 256 and os.environ.get('BLOCK_DENSE', '1') == '1') else 8
tim = ws[off: off + nrec * NS * 8].view(torch.float64).reshape(nrec, NS).cpu().numpy()
blocks = nl[: C // 16 * 16].reshape(-1, 16).max(axis=1)
names = ["rows->LDS", "barrier A", "MFMA", "barrier B", "book", "leap12", "vote", "begin"]
if flow:
    names = ["book: pass", "book: sums, scalars, draw, take", "MFMA", "barriers", "book: U-turn levels, expansion, end / next", "stage12", "vote", "begin"]
elif D <= 256 and os.environ.get("BLOCK_DENSE", "1") == "1":  # the register kernel's phases
    names = ["book: pass", "book: step scalars + draw", "MFMA", "barriers", "book: control + expansion end", "stage12", "vote", "begin/end/draw",
             "book: reductions, energy", "book: proposal copy", "book: U-turn levels", "", "", "", "", ""]
tot = tim.sum(axis=1)
print(f"D={D} C={C} T={T}: {dt*1e3:.3f} ms per transition; leapfrogs/chain/transition mean {nl.mean():.1f}; "
      f"ticks per wave and transition {tot.mean() / T:.0f}")
for k, n in enumerate(names):
    print(f"  {n:16s} {tim[:, k].mean() / T:10.0f} ticks / transition  {100 * tim[:, k].sum() / tot.sum():5.1f} %")
