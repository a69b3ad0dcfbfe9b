#!/usr/bin/env python3
"""fp64 GEMM C[M,N] = A[M,K] B[N,K]^T at mid-size shapes (N = K = D): small-tile kernels (option gemm_small_tiles:
1 auto, 2 = 64x64, 3 = 64x128, 4 = 32x64) against the 128 x 128 paths (0); bitwise equality checked."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd.engine import get_engine
eng = get_engine()
shapes = [tuple(int(v) for v in x.split("x")) for x in sys.argv[1:]] or [(4096, 100), (4096, 200), (4096, 500), (4096, 1000), (1024, 200), (16384, 100), (700, 300), (4096, 201)]
for M, D in shapes:
    A = torch.randn(M, D, dtype=torch.float64, device="cuda")
    B = torch.randn(D, D, dtype=torch.float64, device="cuda")
    eng.set_option("gemm_small_tiles", 0); ref = eng.gemm_nt(A, B)
    line = f"M={M} D={D}:"
    for mode in (0, 1, 2, 3, 4):
        eng.set_option("gemm_small_tiles", mode)
        out = eng.gemm_nt(A, B); torch.cuda.synchronize()
        assert torch.equal(out, ref), f"mode {mode} differs"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): eng.gemm_nt(A, B)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  [{mode}] {us:6.1f} us {2.0*M*D*D/us/1e6:5.1f} TF"
    print(line, flush=True)
eng.set_option("gemm_small_tiles", 1)
