#!/usr/bin/env python3
"""Where a warm-up on the GPU and the restatement's warm-up part: tools/fuzz_adaptation.py's case
70000000547 (D = 7, 5 chains, diagonal Gaussian, sub-wavefront team kernel) flagged chain 4 after 37 steps.  The warm-up is
run for n = 1 .. 37 steps from the same start on both sides; per n: the relative difference of chain 4's (and, for
comparison, chain 0's) position, step size and metric.  The first n with a non-zero difference is where one rounding of a
sum over D = 7 (the team kernel adds in another order than the restatement) enters; what follows is its amplification by
the feedback of the step size into the trajectory.  usage: adapt_divergence_trace.py [case]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
from aehmc_amd import RandomStream, nuts, targets, window_adaptation
from aehmc_amd.engine import get_engine
from oracle import c_oracle as co, np_adaptation as na, np_oracle as no

case = int(sys.argv[1]) if len(sys.argv) > 1 else 70000000547
eng = get_engine()


class OracleNuts:
    def __init__(self, otgt, seed, D, max_exp):
        self.otgt, self.D, self.max_exp = otgt, D, max_exp
        self.rng = co.site_states([seed], 4)
        self.nleap = []

    def __call__(self, state, eps, imm):
        q = np.asarray(state.position, dtype=np.float64).reshape(1, self.D).copy()
        U = np.array([state.potential_energy], dtype=np.float64)
        g = np.asarray(state.potential_energy_grad, dtype=np.float64).reshape(1, self.D).copy()
        res = co.nuts_step(self.otgt, co.Metric(imm, self.D), self.rng, float(eps), q, U, g, max_exp=self.max_exp)
        self.nleap.append(int(res["n_leapfrog"][0]))
        return SimpleNamespace(state=no.IntegratorState(q[0].copy(), None, float(U[0]), g[0].copy()),
                               acceptance_probability=float(res["acceptance_probability"][0]))


# the draws of tools/fuzz_adaptation.py: one(case), in its order
r = np.random.default_rng(case)
kind = r.choice(["diag", "linreg", "std"], p=[0.6, 0.25, 0.15])
full = bool(r.random() < 0.25)
steps = int(r.choice([1, 5, 21, 37, 75, 130, 160]))
C = int(r.choice([1, 2, 5, 9, 33, 70]))
opts = {"resident_nuts": int(r.choice([0, 1, 2])), "resident_min_team": int(r.integers(0, 2))}
assert kind == "diag" and not full, (kind, full)
D = int(r.choice([1, 2, 3, 7, 16, 24, 40, 64, 65, 100, 200, 400, 700]))
mu, sigma = r.normal(size=D), 0.5 + 2 * r.random(D)
tgt = targets.DiagGaussian(mu, sigma)
q0 = mu + sigma * r.normal(size=(C, D))
max_exp = int(np.random.default_rng(case + 2).choice([10, 5]))
seeds = [int(x) for x in np.random.default_rng(case + 1).integers(0, 2 ** 31, size=C)]
otgt = co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
print(f"case {case}: D={D} C={C} steps={steps} max_exp={max_exp} options {opts}")
for k, v in opts.items():
    eng.set_option(k, v)
print(" n | chain 4: |dq|/|q|   |deps|/eps   max|dimm|/imm  leapfrogs(last) | chain 0: |dq|/|q|   |deps|/eps")
try:
    for n in range(1, steps + 1):
        kernel = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
        state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        state, (eps, imm), _ = window_adaptation.run(kernel, state, n)
        pos_g, eps_g, imm_g = state.position.cpu().numpy(), eps.value.cpu().numpy().reshape(-1), imm.value.cpu().numpy().reshape(C, D)
        row = []
        for c in (C - 1, 0):
            Uo, go = no.DiagGaussian(mu, sigma)(q0[c])
            ok = OracleNuts(otgt, seeds[c], D, max_exp)
            st, (eps_o, imm_o) = na.run(ok, no.IntegratorState(q0[c], None, Uo, go), n)
            dq = np.abs(pos_g[c] - st.position).max() / np.abs(st.position).max()
            de = abs(eps_g[c] / eps_o - 1)
            di = np.abs(imm_g[c] / np.asarray(imm_o).reshape(-1) - 1).max()
            row.append((dq, de, di, ok.nleap[-1]))
        print(f"{n:2d} | {row[0][0]:12.3e} {row[0][1]:12.3e} {row[0][2]:12.3e} {row[0][3]:6d}          | {row[1][0]:12.3e} {row[1][1]:12.3e}", flush=True)
finally:
    eng.set_option("resident_nuts", 2)
    eng.set_option("resident_min_team", 0)
