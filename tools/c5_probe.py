import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
C = 1024
rng = np.random.default_rng(0)
N = 100_000
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
tgt = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * rng.normal(size=(C, 2))
kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), tgt)
state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
last, (eps, imm), _ = window_adaptation.run(kernel, state, 1000)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    info, _ = kernel(last, eps, imm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("one kernel call: %.3f ms, max leapfrogs %d" % ((t1 - t0) * 1e3, info.n_leapfrog.max().item()))
    last = info.state._replace(momentum=None)
t0 = time.perf_counter()
s, info, a, d = kernel.sample(last, eps, imm, 20, keep_samples=False)
torch.cuda.synchronize()
print("sample(20): %.3f ms per transition; max total leapfrogs %d" % ((time.perf_counter() - t0) * 1e3 / 20, info.n_leapfrog.max().item()))
