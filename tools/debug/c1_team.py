#!/usr/bin/env python3
"""c1 (README example) per team size: resident_min_team 0 (auto: a whole wavefront) vs 1 (one lane)."""
import os, sys, time
import numpy as np, torch
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
    pos = info.state.position.item()
    return (time.perf_counter() - t0) * 1e6, pos, int(info.n_leapfrog.item())
for mt in (0, 1, 0, 1):
    eng.set_option("resident_min_team", mt)
    for _ in range(10): once()
    r = [once() for _ in range(100)]
    print(f"resident_min_team={mt}: median {np.median([x[0] for x in r]):.1f} us, min {min(x[0] for x in r):.1f} us, position {r[0][1]!r}, leapfrogs {r[0][2]}")
