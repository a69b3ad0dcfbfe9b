import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
D, C = 100, 8
r = np.random.default_rng(D * 1000 + C)
def spd(D):
    A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
mu = torch.as_tensor(r.normal(size=D), device="cuda")
tgt = targets.DenseMVN(mu, torch.as_tensor(P, device="cuda"))
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
eng = get_engine()
E = 8
tr = []
for flow in (1, 0):
    eng.set_option("block_flow", flow)
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=E)
    state = nuts.new_state(q0.clone(), tgt)
    eng._ws = None
    info, upd = kernel(state, 0.3 * D ** -0.25, imm)
    ws = eng._ws
    vec = ((C * D * 8) + 255) & ~255
    off = (26 + 3 * E) * vec
    t = ws[off + 300 * 8: off + 300 * 8 + 8 * 24 * 8].view(torch.float64).reshape(24, 8).cpu().numpy()
    tr.append(t); print("nleap", info.n_leapfrog.cpu().numpy())
np.set_printoptions(linewidth=200, precision=6)
for k in range(22):
    if tr[0][k].any() or tr[1][k].any(): print(k, tr[0][k, :8]); print(k, tr[1][k, :8])
