#!/usr/bin/env python3
"""Debug: c3 at depth 10, both dense modes, per-chain diagnostics saved for offline comparison."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_c3
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
D = 10_000; C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
EPS = 0.5 * D ** -0.25
Sigma, P = build_c3(D, torch.device("cuda"))
tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device="cuda"), P)
seeds = [1000 + c for c in range(C)]
q0 = np.random.default_rng(1234).standard_normal((256, D))[:C]
eng = get_engine()
out = {}
for name, opts in (("lin", dict(dense_linear=1)), ("lit", dict(dense_linear=0)), ("lin_nocompact", dict(dense_linear=1, compact=0)),
                   ("lit_nocompact", dict(dense_linear=0, compact=0)), ("lin_sk0", dict(dense_linear=1, streamk=0))):
    eng.set_option("dense_linear", 1); eng.set_option("compact", 1); eng.set_option("streamk", 2)
    for k, v in opts.items(): eng.set_option(k, v)
    kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=10)
    state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    for t in range(2):
        info, _ = kernel(state, EPS, Sigma)
        state = info.state._replace(momentum=None)
        out[f"{name}_{t}_q8"] = info.state.position[:, :8].cpu().numpy()
        out[f"{name}_{t}_U"] = info.state.potential_energy.cpu().numpy()
        out[f"{name}_{t}_nl"] = info.n_leapfrog.cpu().numpy()
        out[f"{name}_{t}_nd"] = info.num_doublings.cpu().numpy()
        out[f"{name}_{t}_acc"] = info.acceptance_probability.cpu().numpy()
        out[f"{name}_{t}_turn"] = info.is_turning.cpu().numpy()
    print(name, "acc mean", out[f"{name}_0_acc"].mean(), "nl mean", out[f"{name}_0_nl"].mean(), "U[:6]", out[f"{name}_0_U"][:6],
          "nl[:6]", out[f"{name}_0_nl"][:6], "acc[:6]", out[f"{name}_0_acc"][:6])
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/c3_modes.npz", **out)
ref = "lin"
for name in ("lit", "lin_nocompact", "lit_nocompact", "lin_sk0"):
    for t in range(2):
        bad = np.nonzero(np.abs(out[f"{name}_{t}_U"] - out[f"{ref}_{t}_U"]) > 1e-6 * np.abs(out[f"{ref}_{t}_U"]))[0]
        print(name, t, "chains whose U differs from lin:", len(bad), bad[:20])
