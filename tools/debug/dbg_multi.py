import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
eng = get_engine()
D, C, N = 1, 9, 5
r = np.random.default_rng(3 + D)
q0, imm = r.normal(size=(C, D)), 0.5 + r.random(D)
tgt = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
for mt in (1, 0):
    eng.set_option("resident_min_team", mt)
    k1 = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    k2 = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    s1 = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
    samples, info, acc, div = k1.sample(s1, 0.2, imm, N)
    s2 = s1
    for t in range(N):
        i2, _ = k2(s2, 0.2, imm)
        s2 = i2.state._replace(momentum=None)
        d = (samples[t] != i2.state.position).flatten().nonzero().flatten().tolist()
        print(mt, t, "single: chain4 q,U,g,p,acc,nd,turn", i2.state.position[4].item(), i2.state.potential_energy[4].item(), i2.state.potential_energy_grad[4].item(), i2.state.momentum[4].item(), i2.acceptance_probability[4].item(), i2.num_doublings[4].item(), i2.is_turning[4].item())
        print(mt, t, "mismatch chains", d, [(samples[t].flatten()[c].item(), i2.state.position.flatten()[c].item()) for c in d], "nleap", i2.n_leapfrog.flatten().tolist(), "div", i2.is_diverging.flatten().tolist())
