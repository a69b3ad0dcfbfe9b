#!/usr/bin/env python3
"""Vendor fp64 GEMM (torch.matmul -> hipBLASLt / rocBLAS) at the c3 shape, for comparison with
gemm_f64.cuh.  Not used by the product path."""
import sys, torch
N = K = 10000
B = torch.randn(N, K, dtype=torch.float64, device="cuda")
for M in [int(x) for x in sys.argv[1:]] or [4096, 2949, 2048]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda")
    for _ in range(2): (A @ B.T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): C = A @ B.T
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"vendor M={M}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
