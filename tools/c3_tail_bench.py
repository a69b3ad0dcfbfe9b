#!/usr/bin/env python3
"""c3 with a long tail: a few chains get a 10x smaller step size and run to the maximum tree depth
while the rest finish after ~60 leapfrogs -- how much does a lock-step leapfrog cost when only a
handful of rows are left in the chain-batched GEMMs?  usage: python tools/c3_tail_bench.py [n_slow] [C] [D]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_c3
from aehmc_amd import PerChain, RandomStream, nuts, targets

n_slow = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
dev = torch.device("cuda")
Sigma, P = build_c3(D, dev)
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device=dev), P)
eps = np.full(C, 0.5 * D ** -0.25)
eps[:n_slow] *= 0.1
kernel = nuts.new_kernel(RandomStream(seeds=[1000 + c for c in range(C)]), tgt, max_num_expansions=10)
q0 = np.random.default_rng(1234).standard_normal((C, D))
state = nuts.new_state(torch.as_tensor(q0, device=dev), tgt)
e = PerChain(torch.as_tensor(eps, device=dev))
info, _ = kernel(state, e, Sigma)
int(info.n_leapfrog.sum().item())
state = info.state._replace(momentum=None)
torch.cuda.synchronize(); t0 = time.perf_counter()
info, _ = kernel(state, e, Sigma)
nl = info.n_leapfrog.cpu().numpy()
dt = time.perf_counter() - t0
fast, slow = nl[n_slow:], nl[:n_slow]
print(f"C={C} D={D} slow chains {n_slow}: transition {dt:.3f} s; fast chains {fast.mean():.0f} leapfrogs (max {fast.max()}), "
      f"slow chains {slow.mean():.0f} (max {slow.max()}); tail: {(dt) / max(slow.max(), 1) * 1e3:.2f} ms per lock-step leapfrog overall")
