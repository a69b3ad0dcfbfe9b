"""Very many chains of a small target: the first / last chains of a C = 300000 call equal the same chains
(same seeds) run in a small call of their own, for NUTS (sub-wavefront teams, sample()) and HMC."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets
C, D = 300_000, 8
r = np.random.default_rng(0)
mu, sigma, imm = r.normal(size=D), 0.5 + r.random(D), 0.5 + r.random(D)
tgt = targets.DiagGaussian(mu, sigma)
q0 = torch.as_tensor(r.normal(size=(C, D)), device="cuda")
seeds = list(range(C))
for name, mod, extra in (("nuts", nuts, ()), ("hmc", hmc, (9,))):
    k = mod.new_kernel(RandomStream(seeds=seeds), tgt)
    s = mod.new_state(q0, tgt)
    samples, info, acc, div = k.sample(s, 0.3, imm, *extra, 3)
    for sl in (slice(0, 40), slice(C - 33, C)):
        k2 = mod.new_kernel(RandomStream(seeds=seeds[sl]), tgt)
        s2 = mod.new_state(q0[sl].clone(), tgt)
        sm2, i2, a2, d2 = k2.sample(s2, 0.3, imm, *extra, 3)
        assert torch.equal(sm2, samples[:, sl]), name
        # (the team size -- lanes per chain -- follows the chain count: sums over D in another order, last bits)
        assert torch.allclose(a2, acc[:, sl], rtol=1e-10, atol=0) and torch.equal(i2.n_leapfrog, info.n_leapfrog[sl]), name
    print(name, "ok: mean acceptance", acc.mean().item(), "leapfrogs", int(info.n_leapfrog.sum().item()))
