#!/usr/bin/env python3
"""leapfrog/s of bench.py's secondary diag-nuts / custom workloads on k_nuts_wide, a few transitions each.
usage: diag_nuts_rate.py [D ...]   (4096 chains, isotropic Gaussian, diagonal mass, NUTS depth 10)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import diag_case
dims = [int(x) for x in sys.argv[1:]] or [10_000, 5_000, 2_000, 1_000]
for D in dims:
    state, step = diag_case("nuts", D, 4096, torch.device("cuda"))
    info, _ = step(state)
    state = info.state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nl = 0
    infos = []
    for _ in range(4):
        info, _ = step(state)
        state = info.state._replace(momentum=None)
        infos.append(info.n_leapfrog)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nl = sum(int(x.sum()) for x in infos)
    print(f"diag-nuts D={D}: {nl / dt:.4e} leapfrog/s ({nl / 4 / 4096:.1f} leapfrogs/chain, {dt / 4 * 1e3:.1f} ms/transition)", flush=True)
