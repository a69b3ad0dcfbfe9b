"""Where does the host time of one c2 bench step go?  (cProfile over 50 steps; wall vs kernel.)"""
import cProfile, pstats, time, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aehmc_amd import RandomStream, hmc, targets
from aehmc_amd.engine import get_engine
C, D, L, NT = 4096, 100, 32, 100
dev = torch.device("cuda")
q0 = np.random.default_rng(1234).standard_normal((C, D))
tgt = targets.IsoGaussian()
imm = torch.ones(D, dtype=torch.float64, device=dev)
kernel = hmc.new_kernel(RandomStream(seeds=[1000 + c for c in range(C)]), tgt)
state = hmc.new_state(torch.as_tensor(q0, device=dev), tgt)
def step(st):
    return kernel.sample(st, 0.1, imm, L, NT, keep_samples=False)[1]
for _ in range(3):
    state = step(state).state._replace(momentum=None)
torch.cuda.synchronize()
for sync in (False, True):
    t0 = time.perf_counter()
    for _ in range(50):
        info = step(state); state = info.state._replace(momentum=None)
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("sync each step" if sync else "async", (time.perf_counter() - t0) / 50 * 1e3, "ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    info = step(state); state = info.state._replace(momentum=None)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
