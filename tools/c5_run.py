#!/usr/bin/env python3
"""bench.py's c5 workload as two launches of k_nuts_linreg (warm-up, then sample(T)) for rocprofv3: the
LAST dispatch of the kernel is the timed sample() launch.  usage: c5_run.py [C] [warmup] [T]; prints one JSON line
with the leapfrog total of the sample launch (the run is seeded: every profiler pass sees the same launch)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rng = np.random.default_rng(0)
N = 100_000
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
target = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * np.random.default_rng(1).normal(size=(C, 2))
kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), target)
state = nuts.new_state(torch.as_tensor(q0, device="cuda"), target)
state, (eps, imm), _ = window_adaptation.run(kernel, state, W)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, info, _, _ = kernel.sample(state, eps, imm, T, keep_samples=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
nl = info.n_leapfrog.cpu().numpy()
print(json.dumps({"chains": C, "rows": N, "warmup": W, "transitions": T, "leapfrogs": int(nl.sum()),
                  "leapfrogs_max_chain": int(nl.max()), "wall_ms": dt * 1e3, "leapfrogs_per_s": float(nl.sum() / dt)}))
