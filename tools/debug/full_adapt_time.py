#!/usr/bin/env python3
"""window_adaptation.run(is_mass_matrix_full=True) + sampling with the adapted per-chain dense matrices at small D:
wall time with the small-dense single-launch kernel against the lock-step path (resident_nuts = 0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import PerChain, RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W = int(sys.argv[3]) if len(sys.argv) > 3 else 300
r = np.random.default_rng(0)
A = r.normal(size=(D, D)); P = A @ A.T / D + np.eye(D); P = 0.5 * (P + P.T)
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
eng = get_engine()
MODES = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [2, 0, 2, 0]
for mode in MODES:
    eng.set_option("resident_nuts", mode)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, (eps, imm), _ = window_adaptation.run(kernel, state, num_steps=W, is_mass_matrix_full=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    samples, info = kernel.sample(state, eps, imm, 100)[:2]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    e = eps.value if isinstance(eps, PerChain) else eps
    print(f"resident_nuts={mode} D={D} C={C}: warm-up {W} steps {t1-t0:.3f} s, 100 samples {t2-t1:.3f} s, "
          f"median eps {float(torch.as_tensor(e).median()):.3f}", flush=True)
