#!/usr/bin/env python3
"""NUTS on the linear-regression target with a fixed step size (no adaptation): time per
lock-step leapfrog of C chains.  usage: python tools/linreg_nuts_bench.py [C] [eps] [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
rng = np.random.default_rng(0)
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
tgt = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * np.random.default_rng(1).normal(size=(C, 2))
imm = torch.ones(2, dtype=torch.float64, device="cuda")
eng = get_engine()
for resident in (2, 0):
    eng.set_option("resident_nuts", resident)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=7)
    state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    info, _ = kernel(state, eps, imm)
    int(info.n_leapfrog.sum().item())  # load every op of the timed loop first
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    tot = 0
    for _ in range(reps):
        info, _ = kernel(info.state._replace(momentum=None), eps, imm)
        tot += int(info.n_leapfrog.sum().item())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nl = info.n_leapfrog.double()
    print(f"C={C} N={N} eps={eps} resident={resident}: {tot/dt:.3e} leapfrog/s  {dt/reps*1e3:.2f} ms/transition  "
          f"leapfrogs/chain mean {nl.mean().item():.1f} max {nl.max().item():.0f}  "
          f"-> {dt/reps/nl.max().item()*1e6:.1f} us per lock-step leapfrog, {tot/dt*N:.3e} rows/s")
eng.set_option("resident_nuts", 2)
