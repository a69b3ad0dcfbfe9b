#!/usr/bin/env python3
"""A few launches of the fp64 MFMA GEMM at one shape (for counter collection). usage: gemm_once.py M [streamk]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd.engine import get_engine
eng = get_engine()
M = int(sys.argv[1]); eng.set_option("streamk", int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = K = 10000
B = torch.randn(N, K, dtype=torch.float64, device="cuda")
A = torch.randn(M, K, dtype=torch.float64, device="cuda")
for _ in range(3): eng.gemm_nt(A, B)
torch.cuda.synchronize()
