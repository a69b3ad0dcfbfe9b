#!/usr/bin/env python3
"""Where does a single-chain NUTS transition (config c1) spend its time on the host?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
eng = get_engine()
target = targets.StdNormal()
def once():
    kernel = nuts.new_kernel(RandomStream(seed=0), target)
    state = nuts.new_state(0.0, target)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, _ = kernel(state, 1e-2, 1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    pos = info.state.position.item()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, pos
for _ in range(5): once()
r = [once() for _ in range(50)]
import numpy as np
a = np.array([x[:3] for x in r])
print("enqueue (python + C-ABI) %.1f us, wait for kernel %.1f us, item %.1f us; position %r" % (*np.median(a, axis=0), r[0][3]))
import cProfile, pstats
kernel = nuts.new_kernel(RandomStream(seed=0), target)
state = nuts.new_state(0.0, target)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    info, _ = kernel(state, 1e-2, 1.0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
