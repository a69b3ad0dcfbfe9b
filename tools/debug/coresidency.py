#!/usr/bin/env python3
"""Does a streaming kernel execute BESIDE the persistent fp64 GEMM (one 256-thread workgroup per CU, 368 of 512
registers per SIMD, 110 KB LDS)?  GEMM on one stream, a torch elementwise add on another: alone, and together."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd.engine import get_engine
eng = get_engine()
dev = eng.device
M, D = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 10_000
A = torch.randn(M, D, dtype=torch.float64, device=dev)
B = torch.randn(D, D, dtype=torch.float64, device=dev)
x = torch.randn(M * D * 2, dtype=torch.float64, device=dev)
y = torch.randn(M * D * 2, dtype=torch.float64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def gemm():
    with torch.cuda.stream(s1):
        eng.gemm_nt(A, B)
def adds(k=4):
    with torch.cuda.stream(s2):
        for _ in range(k):
            x.add_(y)
def both():
    gemm(); adds()
def both_rev():
    adds(); gemm()
print(f"M={M}: gemm alone {t(gemm):.3f} ms; 4 adds alone {t(adds):.3f} ms; gemm then adds (2 streams) {t(both):.3f} ms; adds then gemm {t(both_rev):.3f} ms")
