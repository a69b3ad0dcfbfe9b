#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, targets
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
X = rng.normal(0, 1, size=(10_000,)); y = 3 * X + rng.normal(0, 1)
tgt = targets.LinearRegression(X, y)
q0 = np.tile(np.array([3.0, np.log(0.21)]), (C, 1))
kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
state = hmc.new_state(torch.as_tensor(q0, device="cuda"), tgt)
for _ in range(2):
    info, _ = kernel(state, 5e-5, np.array([1.0, 1.0]), 1024)
torch.cuda.synchronize()
