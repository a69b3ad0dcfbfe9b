"""c5 tree statistics: per-transition mean / max leapfrog count over the chains of one GPU after warm-up,
and the time of a fused sample() call (what bench.py --config c5 times)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets, window_adaptation

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda")
rng = np.random.default_rng(0)
N = 100_000
X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
target = targets.LinearRegression(X, y)
q0 = np.array([3.0, np.log(0.5)]) + 0.05 * np.random.default_rng(1).normal(size=(C, 2))
kernel = nuts.new_kernel(RandomStream(seeds=[5000 + c for c in range(C)]), target)
state = nuts.new_state(torch.as_tensor(q0, device=dev), target)
state, (eps, imm), _ = window_adaptation.run(kernel, state, W)
torch.cuda.synchronize()
print("eps", eps.min().item() if hasattr(eps, "min") else eps, "imm", imm[:2] if hasattr(imm, "__getitem__") else imm)
tot = 0; mx = 0
for t in range(20):
    info, _ = kernel(state, eps, imm)
    state = info.state._replace(momentum=None)
    nl = info.n_leapfrog.cpu().numpy()
    g4 = nl.reshape(-1, 4).max(1)
    print(t, "mean", nl.mean(), "max", nl.max(), "mean of 4-group max", g4.mean(), "hist", np.bincount(nl)[:40].tolist())
for n in (10, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, info, _, _ = kernel.sample(state, eps, imm, n, keep_samples=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(n, "transitions", dt / n * 1e3, "ms each;", int(info.n_leapfrog.sum().item()) / dt, "leapfrog/s")
