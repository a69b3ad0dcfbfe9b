#!/usr/bin/env python3
"""c1 (README example: one chain, one NUTS transition from a fresh kernel): where the 0.28 ms go (host profile)."""
import cProfile, pstats, time, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
target = targets.StdNormal()
def one():
    kernel = nuts.new_kernel(RandomStream(seed=0), target)
    state = nuts.new_state(0.0, target)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, _ = kernel(state, 1e-2, 1.0)
    t1 = time.perf_counter()
    pos = info.state.position.item()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, pos
for _ in range(20): one()
N = 200
a = b = 0.0
for _ in range(N):
    x, y, pos = one(); a += x; b += y
print(f"call returns after {a/N*1e6:.1f} us, .item() after another {b/N*1e6:.1f} us; position {pos!r}")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): one()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(30)
