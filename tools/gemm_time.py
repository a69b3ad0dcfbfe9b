#!/usr/bin/env python3
"""Timing of the fp64 MFMA GEMM at the c3 shape for several live-row counts, stream-K modes 2 (128x256 tiles) / 1 (128x128) / 0 (off),
interleaved in one process; every mode is checked against mode 0 bit for bit.  usage: python tools/gemm_time.py [M ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd.engine import get_engine
eng = get_engine()
N = K = 10000
B = torch.randn(N, K, dtype=torch.float64, device="cuda")
for M in [int(x) for x in sys.argv[1:]] or [4096, 2900, 2048, 1024, 512]:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda")
    eng.set_option("streamk", 0); ref = eng.gemm_nt(A, B)
    for sk in (2, 1, 0, 2, 1, 0):
        eng.set_option("streamk", sk)
        out = eng.gemm_nt(A, B); torch.cuda.synchronize()
        assert os.environ.get("AEHMC_NOCHECK") or torch.equal(out, ref), f"mode {sk} differs from the one-tile-per-workgroup kernel"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.gemm_nt(A, B)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M={M} streamk={sk}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
eng.set_option("streamk", 2)
