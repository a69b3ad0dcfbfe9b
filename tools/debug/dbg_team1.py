import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
from oracle import c_oracle as co
eng = get_engine()
for D, C, mt in [(1, 64, 1), (1, 300, 1)]:
    r = np.random.default_rng(3 + D)
    q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
    mu, sigma = r.normal(size=D), 0.5 + r.random(D)
    tgt, otgt = targets.DiagGaussian(mu, sigma), co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    eng.set_option("resident_min_team", mt)
    k = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    s = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    rng = co.site_states(list(range(C)), 4)
    q, U, g = co.new_state(otgt, q0.copy())
    for t in range(6):
        info, upd = k(s, 0.2, imm)
        s = info.state._replace(momentum=None)
        res = co.nuts_step(otgt, co.Metric(imm, D), rng, 0.2, q, U, g)
        bad = np.nonzero(info.n_leapfrog.cpu().numpy().reshape(-1) != res["n_leapfrog"])[0]
        rbad = np.nonzero((upd[list(upd)[0]].cpu().numpy().view(np.uint64)[:, :, :2] != rng[:, :, :2]).any(axis=(1, 2)))[0]
        print(D, C, mt, t, "chains with wrong n_leapfrog:", bad.tolist()[:10], "gpu", info.n_leapfrog.cpu().numpy().reshape(-1)[bad][:10].tolist(),
              "oracle", res["n_leapfrog"][bad][:10].tolist(), "rng mismatch:", rbad.tolist()[:10])
        gr = upd[list(upd)[0]].cpu().numpy().view(np.uint64)
        for c in rbad[:3]:
            print("   chain", c, "sites differing:", [k for k in range(4) if (gr[c, k, :2] != rng[c, k, :2]).any()], "gpu", gr[c].tolist(), "oracle", rng[c].tolist())
        if len(bad):
            # resync the GPU state to the oracle's so that later transitions are comparable
            break
eng.set_option("resident_min_team", 0)
