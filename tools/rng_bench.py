#!/usr/bin/env python3
"""Device RNG throughput: normals and Bernoulli draws per second (one PCG64 stream per chain).
usage: python tools/rng_bench.py [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream
from aehmc_amd.engine import get_engine

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = get_engine()
rng = torch.as_tensor(RandomStream(seeds=list(range(C))).sites(1).astype(np.int64).reshape(C, 4), device="cuda")
rng = rng.contiguous()  # [C, 4]: one call site per chain
def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for n in (64, 100, 1000, 10000):
    dt = timeit(lambda: eng.rng_normals(rng, n))
    print(f"normals   C={C} n={n:6d}: {dt*1e6:9.1f} us/call  {C*n/dt:.3e} normals/s")
p = torch.rand(C, 100, dtype=torch.float64, device="cuda")
dt = timeit(lambda: eng.rng_bernoulli(rng, p))
print(f"bernoulli C={C} n=   100: {dt*1e6:9.1f} us/call  {C*100/dt:.3e} draws/s")
