#!/usr/bin/env python3
"""Would ONE GEMM with the stacked operand [P; imm P] (N = 2 D) beat the two dependent GEMMs of a dense-metric
leapfrog (g' = P r, then w' = imm g')?  Timing only (HIP events through the engine's profile hooks)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd.engine import get_engine
eng = get_engine()
dev = eng.device
D = 10_000
B1 = torch.randn(D, D, dtype=torch.float64, device=dev)
B2 = torch.randn(2 * D, D, dtype=torch.float64, device=dev)
for M in (4096, 2949, 2048, 1200):
    A = torch.randn(M, D, dtype=torch.float64, device=dev)
    def pair():
        g = eng.gemm_nt(A, B1)
        eng.gemm_nt(g, B1)
    def one():
        eng.gemm_nt(A, B2)
    res = []
    for fn in (pair, one, pair, one):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 8 * 1e3)
    print(f"M={M}: two GEMMs N=1e4: {res[0]:.3f} / {res[2]:.3f} ms; one GEMM N=2e4: {res[1]:.3f} / {res[3]:.3f} ms; ratio {res[3]/res[2]:.4f}")
