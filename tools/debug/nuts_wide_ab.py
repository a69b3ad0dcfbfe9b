#!/usr/bin/env python3
"""bench.py's diag-nuts workload, bench-style timing (no sync inside the loop): usage nuts_wide_ab.py [D] [transitions]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import diag_case
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
state, step = diag_case("nuts", D, 4096, torch.device("cuda"))
for _ in range(3):
    info, _ = step(state); state = info.state._replace(momentum=None)
torch.cuda.synchronize(); t0 = time.perf_counter(); nl = torch.zeros((), dtype=torch.int64, device="cuda")
for _ in range(n):
    info, _ = step(state); state = info.state._replace(momentum=None); nl += info.n_leapfrog.sum()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("D=%d: %.3e leapfrog/s %.2f ms/transition" % (D, int(nl.item()) / dt, dt / n * 1e3))
