#!/usr/bin/env python3
"""window_adaptation.run(is_mass_matrix_full=True) at MID-size D (per-chain D x D metrics, 64 < D <= 512: Welford in
global memory, one wavefront per matrix at window ends) + sampling with the adapted matrices, coordinate-wise target
with unequal scales: wall time with csrc/nuts_pc_dense.cuh (one launch per transition / per sample() call) against the
lock-step path (pc_dense = 0).  usage: full_adapt_mid.py [D] [C] [warm-up steps] [samples]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import PerChain, RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W = int(sys.argv[3]) if len(sys.argv) > 3 else 300
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
r = np.random.default_rng(0)
tgt = targets.DiagGaussian(r.normal(size=D), np.exp(r.normal(size=D)))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
eng = get_engine()
for pc in (1, 0):
    eng.set_option("pc_dense", pc)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, (eps, imm), _ = window_adaptation.run(kernel, state, num_steps=W, is_mass_matrix_full=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    samples, info = kernel.sample(state, eps, imm, N)[:2]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    e = eps.value if isinstance(eps, PerChain) else eps
    nl = float(info.n_leapfrog.double().sum())
    print(f"pc_dense={pc} D={D} C={C}: warm-up {W} steps {t1 - t0:.3f} s, {N} samples {t2 - t1:.3f} s "
          f"({nl / C / N:.1f} leapfrogs per transition, {(nl + 3.0 * C * N) * D * D * 8 / (t2 - t1) / 1e12:.2f} TB/s on the matrix bytes), "
          f"median eps {float(torch.as_tensor(e).median()):.3f}", flush=True)
