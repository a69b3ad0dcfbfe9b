#!/usr/bin/env python3
"""Block-resident dense NUTS (k_nuts_block_flow) against the lock-step path over a grid of (D, C, T), each case in its
own process: prints ok / MISMATCH / CRASH.  usage: flow_grid.py [case D C T]"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def case(D, C, T, td):
    import torch
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import get_engine
    r = np.random.default_rng(D * 1000 + C)
    def spd(D):
        A = r.normal(size=(D, D)); M = A @ A.T / D + np.eye(D); return 0.5 * (M + M.T)
    P, imm = spd(D), torch.as_tensor(spd(D), device="cuda")
    mu = torch.as_tensor(r.normal(size=D), device="cuda")
    tgt = targets.DenseMVN(mu, torch.as_tensor(P, device="cuda")) if td else targets.DiagGaussian(mu, torch.as_tensor(0.5 + r.random(D), device="cuda"))
    q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
    eng = get_engine()
    outs = []
    for blk in (1, 0):
        eng.set_option("block_dense", blk)
        eng.set_option("block_roll", int(os.environ.get("DBG", "0")))
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=8)
        state = nuts.new_state(q0.clone(), tgt)
        if T == 1:
            info, upd = kernel(state, 0.3 * D ** -0.25, imm)
            outs.append((info.state.position.cpu().numpy(), info.n_leapfrog.cpu().numpy(), info.acceptance_probability.cpu().numpy()))
        else:
            samples, info = kernel.sample(state, 0.3 * D ** -0.25, imm, T)[:2]
            outs.append((samples.cpu().numpy(), info.n_leapfrog.cpu().numpy(), info.acceptance_probability.cpu().numpy()))
    same = all(np.array_equal(x, y) for x, y in zip(*outs))
    if not same:
        nl = [np.flatnonzero(outs[0][1] != outs[1][1]).tolist()[:40], outs[0][1][:16].tolist(), outs[1][1][:16].tolist()]
        qq = outs[0][0].reshape(-1, C, D)[0], outs[1][0].reshape(-1, C, D)[0]
        bad = np.flatnonzero(np.any(qq[0] != qq[1], axis=1)).tolist()[:40]
        print("MISMATCH nleap-diff chains", nl, "position-diff chains (first transition)", bad, flush=True)
    else:
        print("ok", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "case":
    case(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    sys.exit(0)
grid = [(100, 8, 1, 1)]
if os.environ.get("GRID") == "full":
    grid = [(100, 16, 1, 1), (100, 16, 3, 1), (100, 5, 1, 1), (100, 64, 1, 1), (100, 48, 2, 1), (100, 1024, 2, 1), (70, 16, 2, 1), (100, 16, 2, 0),
            (128, 33, 2, 1), (130, 16, 2, 1), (200, 19, 2, 1), (256, 19, 2, 1), (256, 19, 1, 0), (100, 4096, 4, 1)]
for g in grid:
    p = subprocess.run([sys.executable, __file__, "case"] + [str(x) for x in g], capture_output=True, text=True)
    res = p.stdout.strip().split("\n")[-1] if p.returncode == 0 else f"CRASH rc={p.returncode} " + p.stderr.strip().split("\n")[0][:100]
    print(g, res, flush=True)
