#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_nuts_wide (needs `make -C aehmc_amd/csrc timing`;
run with AEHMC_AMD_LIB=aehmc_amd/libaehmc_hip_timing.so).  usage: wide_phases.py [D] [C]"""
import ctypes as ct, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eps = 0.5 * D ** -0.25
eng = get_engine()
q0 = torch.as_tensor(np.random.default_rng(0).standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
tgt = targets.IsoGaussian()
kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = nuts.new_state(q0, tgt)
for _ in range(2):
    info, _ = kernel(state, eps, imm)
    state = info.state._replace(momentum=None)
torch.cuda.synchronize()
# linreg_part sits in the workspace right after the work vectors: find it through the layout
ws = eng._ws
ld = (D + 511) // 512 * 512 if D > 512 else D
vec = ((C * ld * 8) + 255) & ~255
n_vec = 3 + 6 + 6 + 2 + 2 * 10 + 3
off = n_vec * vec
tim = ws[off: off + C * 8 * 8].view(torch.float64).reshape(C, 8).cpu().numpy()
nl = info.n_leapfrog.cpu().numpy()
names = ["pass", "prefetch", "reduce", "scalars", "levels", "take", "expansion", "loop"]
tot = tim.sum(axis=1)
print(f"D={D} C={C}: leapfrogs/chain {nl.mean():.1f}; cycles per leapfrog (mean over chains) total {np.mean(tot / nl):.0f}")
for k, n in enumerate(names):
    print(f"  {n:10s} {np.mean(tim[:, k] / nl):9.0f} cycles/leapfrog  {100 * tim[:, k].sum() / tot.sum():5.1f} %")
