import os, sys, torch
sys.path.insert(0, "/root/repo")
from aehmc_amd.engine import get_engine
eng = get_engine()
N = K = 10000
B = torch.randn(N, K, dtype=torch.float64, device="cuda")
for M in (4096, 2900):
    A = torch.randn(M, K, dtype=torch.float64, device="cuda")
    eng.gemm_nt(A, B); torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.gemm_nt(A, B)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{os.environ.get('TAG','')} M={M}: {best:.3f} ms {2.0*M*N*K/best/1e9:.1f} TF")
