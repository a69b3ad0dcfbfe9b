#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_nuts_linreg (needs `make -C aehmc_amd/csrc timing`;
run with AEHMC_AMD_LIB=aehmc_amd/libaehmc_hip_timing.so).  usage: linreg_phases.py [C] [transitions]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(0)
N = 100_000
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
target = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * np.random.default_rng(1).normal(size=(C, 2))
kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), target)
state = nuts.new_state(torch.as_tensor(q0, device="cuda"), target)
state, (eps, imm), _ = window_adaptation.run(kernel, state, 300)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, info, _, _ = kernel.sample(state, eps, imm, T, keep_samples=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
eng = get_engine()
ws = eng._ws
vec = ((C * 2 * 8) + 255) & ~255
off = 17 * vec  # ws_layout: cur 3, ends 6, slots 6, psum, psub, then ckp
tim = ws[off: off + C * 8 * 8].view(torch.float64).reshape(C, 8).cpu().numpy()
nl = info.n_leapfrog.cpu().numpy()
print(f"{T} transitions of {C} chains: {dt*1e3:.2f} ms, {nl.sum()/dt:.3e} leapfrog/s; leapfrogs/chain mean {nl.mean():.1f} max {nl.max()}")
names = ["half+publish+barrier", "sweep", "barrier", "target finish", "tree", "transition end/begin", "-", "-"]
tot = tim.sum(axis=1)
print(f"ticks per chain: mean {tot.mean():.0f} (= {dt*1e6:.0f} us wall => {tot.mean()/dt/1e6:.1f} ticks/us)")
for k, n in enumerate(names[:6]):
    print(f"  {n:22s} {np.mean(tim[:, k] / nl):9.1f} ticks/leapfrog  {100 * tim[:, k].sum() / tot.sum():5.1f} %")
off2 = off + 10 * vec  # cks follows ckp (max_num_expansions = 10 levels)
tg = ws[off2: off2 + C * 8 * 8].view(torch.float64).reshape(C, 8).cpu().numpy()
print("row-serving waves: sweep %.1f, barrier after sweep %.1f, barrier before %.1f ticks per round (chain waves: %.1f / %.1f / %.1f incl. their serial work)"
      % tuple(np.mean(x) / (tot.mean() / np.mean(tim.sum(axis=1) / 1)) * 0 + np.mean(x) for x in (tg[:, 1], tg[:, 2], tg[:, 0], tim[:, 1], tim[:, 2], tim[:, 0])))
