#!/usr/bin/env python3
"""Config c5 at one GPU's share: regression (1e5 rows), D=2, NUTS, window adaptation.
usage: python tools/c5_bench.py [chains] [warmup_steps] [sample_steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(0)
N = 100_000
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
tgt = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * rng.normal(size=(C, 2))
kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), tgt)
state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
torch.cuda.synchronize(); t0 = time.perf_counter()
last, (eps, imm), _ = window_adaptation.run(kernel, state, W)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"warm-up: {W} steps in {t1-t0:.2f} s ({(t1-t0)/W*1e3:.1f} ms/step); median eps {eps.value.median().item():.3g}")
samples, info, acc, div = kernel.sample(last, eps, imm, S, keep_samples=False)
torch.cuda.synchronize(); t2 = time.perf_counter()
nl = int(info.n_leapfrog.sum().item())
print(f"sampling: {S} transitions in {t2-t1:.2f} s; {nl/(t2-t1):.3e} leapfrog/s; {nl/S/C:.1f} leapfrogs/chain/transition; "
      f"data rows touched/s {nl/(t2-t1)*N:.3e}")
nl = info.n_leapfrog.cpu().numpy()
import numpy as np
print("per-chain leapfrog totals over", S, "transitions: mean", nl.mean(), "median", np.median(nl), "p99", np.percentile(nl, 99), "max", nl.max())
info1, _ = kernel(info.state._replace(momentum=None), eps, imm)
n1 = info1.n_leapfrog.cpu().numpy()
print("one transition: mean", n1.mean(), "max", n1.max(), "hist", np.bincount(np.minimum(np.log2(np.maximum(n1,1)).astype(int), 10)))
print("eps quantiles", np.percentile(eps.value.cpu().numpy(), [0, 1, 50, 99, 100]))
