#!/usr/bin/env python3
"""Debug: fp64 GEMM vs torch.matmul over row counts, N = K = 1e4 (c3's shapes)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd.engine import get_engine
eng = get_engine()
N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
g = torch.Generator(device="cuda").manual_seed(0)
B = torch.randn(N, K, dtype=torch.float64, device="cuda", generator=g)
Afull = torch.randn(4096, K, dtype=torch.float64, device="cuda", generator=g)
for M in (1, 64, 128, 129, 200, 256, 257, 300, 384, 385, 512, 640, 768, 769, 1000, 1100, 2048, 2949, 4096):
    A = Afull[:M].contiguous()
    ref = A @ B.T
    res = {}
    for sk in (2, 1, 0):
        eng.set_option("streamk", sk)
        out = eng.gemm_nt(A, B)
        torch.cuda.synchronize()
        err = (out - ref).abs().max().item()
        badrows = ((out - ref).abs().amax(dim=1) > 1e-8).nonzero().flatten()
        res[sk] = out
        print(f"M={M} streamk={sk}: max err {err:.3e} bad rows {badrows.numel()} {badrows[:6].tolist()} {badrows[-3:].tolist()}", flush=True)
    print("   bitwise 2==0:", torch.equal(res[2], res[0]), " 1==0:", torch.equal(res[1], res[0]))
eng.set_option("streamk", 2)
