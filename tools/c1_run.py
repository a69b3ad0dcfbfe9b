#!/usr/bin/env python3
"""Fixed workload for rocprofv3: BASELINE config c1 (README example: 1-D standard normal, NUTS, step size 1e-2, one
chain, RandomStream(seed=0)) N times -- every launch is the same 136-leapfrog transition.  usage: c1_run.py [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, nuts, targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
target = targets.StdNormal()
pos = nl = None
for _ in range(N):
    kernel = nuts.new_kernel(RandomStream(seed=0), target)
    info, _ = kernel(nuts.new_state(0.0, target), 1e-2, 1.0)
    pos, nl = info.state.position.item(), int(info.n_leapfrog.item())
torch.cuda.synchronize()
print(f"c1: position {pos!r} after {nl} leapfrogs, {N} launches")
