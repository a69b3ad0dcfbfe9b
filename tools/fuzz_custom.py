#!/usr/bin/env python3
"""Randomised parity sweep for USER-DEFINED targets (run-time compiled kernels): random sampler, target form
(coordinate-wise Student-t given by its density / joint AR(1) density / logistic regression given by its densities), D, metric, chain count and engine options against
the numpy restatement (oracle/np_oracle.py) with the analytic gradient -- positions, energies and gradients at 1e-9,
leapfrog counts / doublings / flags exact.  Every kernel family a user target can reach is drawn: register-resident,
workgroup-per-chain, block-resident, lock-step, the joint one-launch kernels below and above 64 coordinates.
usage: fuzz_custom.py [seconds] [seed]   (FUZZ_COUNT=N or FUZZ_CASES=a,b,... : fixed case lists)"""
import os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aehmc_amd import RandomStream, hmc, nuts, targets
from aehmc_amd.engine import get_engine
from oracle import np_oracle as no

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = get_engine()
RTOL = 1e-9
STUDENT_T = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const double nu = prm[0][i], s = prm[1][i];
  const T z = q / s;
  return -0.5 * (nu + 1.0) * log1p(z * z / nu);
}
"""
AR1 = """
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  const double rho = 0.6, s2 = 1.0 - rho * rho;
  auto lp = -0.5 * q[0] * q[0];
  for (int i = 1; i < q.size(); i++) {
    auto d = q[i] - rho * q[i - 1];
    lp += -0.5 * d * d / s2;
  }
  return lp;
}
"""


class StudentT:
    def __init__(self, nu, s):
        self.nu, self.s = nu, s

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        z = q / self.s
        u = 0.0
        for t in 0.5 * (self.nu + 1.0) * np.log1p(z * z / self.nu):
            u += t
        return float(u), (self.nu + 1.0) * z / (self.nu + z * z) / self.s


GLM = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / (prm[0][0] * prm[0][0]); }
"""


class Logistic:
    def __init__(self, X, y, tau):
        self.X, self.y, self.tau = X, y, tau

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        z = self.X @ q
        loss = np.where(z > 0, z + np.log1p(np.exp(-np.abs(z))), np.log1p(np.exp(-np.abs(z)))) - self.y * z
        d = 1.0 / (1.0 + np.exp(-z)) - self.y
        return float(loss.sum() + (0.5 * q * q / self.tau ** 2).sum()), self.X.T @ d + q / self.tau ** 2


class Ar1:
    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        rho, s2 = 0.6, 1.0 - 0.36
        lp = -0.5 * q[0] * q[0]
        d = q[1:] - rho * q[:-1]
        for x in d:
            lp += -0.5 * x * x / s2
        g = np.zeros_like(q)
        g[0] = -q[0]
        g[1:] += -d / s2
        g[:-1] += rho * d / s2
        return float(-lp), -g


def one(case):
    r = np.random.default_rng(case)
    sampler = r.choice(["nuts", "hmc"])
    form = r.choice(["elem", "joint", "glm"], p=[0.45, 0.3, 0.25])
    mk = r.choice(["diag", "dense"], p=[0.7, 0.3])
    if form == "elem":
        D = int(r.choice([3, 40, 70, 130, 200, 300, 600, 1100, 2500] if mk == "diag" else [5, 40, 70, 130, 200, 300]))
    elif form == "glm":
        D = int(r.choice([1, 2, 7, 8, 9, 16, 17, 31, 32, 33, 45]))
    else:
        D = int(r.choice([2, 9, 33, 64, 65, 100, 150, 192, 193, 260]))
    C = int(r.choice([1, 3, 5, 17]))
    if form == "glm" and r.random() < 0.15:
        C = 1030  # (above the chain count up to which D > 16 takes the one-launch kernels)
    opts = {"resident_nuts": int(r.choice([0, 2], p=[0.25, 0.75])), "fused_hmc": int(r.choice([0, 1], p=[0.25, 0.75])),
            "block_dense": int(r.choice([0, 1, 2], p=[0.2, 0.6, 0.2]))}
    if form == "elem":
        nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
        tgt, otgt = targets.Custom(STUDENT_T, params=[nu, s]), StudentT(nu, s)
    elif form == "glm":
        N = int(r.choice([1, 5, 63, 64, 65, 200, 1000, 2500]))
        X = r.normal(size=(N, D)) / np.sqrt(D)
        y = (r.random(N) < 0.5).astype(np.float64)
        tgt, otgt = targets.CustomGLM(GLM, X, y, params=[[2.0]]), Logistic(X, y, 2.0)
    else:
        tgt, otgt = targets.CustomJoint(AR1, dim=D), Ar1()
    if mk == "diag":
        imm = 0.5 + r.random(D)
        immg = imm
    else:
        A = r.normal(size=(D, D))
        imm = A @ A.T / D + np.eye(D)
        imm = 0.5 * (imm + imm.T)
        immg = torch.as_tensor(imm, device="cuda")
    q0 = 0.5 * r.normal(size=(C, D))
    eps = float(r.choice([0.05, 0.15, 0.3]))
    seeds = [int(x) for x in r.integers(0, 2 ** 31, size=C)]
    max_exp, L, n = int(r.choice([3, 5])), int(r.choice([1, 4, 9])), 2
    for k, v in opts.items():
        eng.set_option(k, v)
    try:
        if sampler == "nuts":
            kern = nuts.new_kernel(RandomStream(seeds=seeds), tgt, max_num_expansions=max_exp)
            okern = {c: no.nuts_kernel(no.RandomStream(seeds[c]), otgt, max_num_expansions=max_exp) for c in (range(C) if C < 100 else (0, 1, C // 2, C - 1))}
            extra = ()
        else:
            kern = hmc.new_kernel(RandomStream(seeds=seeds), tgt)
            okern = {c: no.hmc_kernel(no.RandomStream(seeds[c]), otgt) for c in (range(C) if C < 100 else (0, 1, C // 2, C - 1))}
            extra = (L,)
        state = nuts.new_state(torch.as_tensor(q0, device="cuda"), tgt)
        ostate = {c: no.new_state(q0[c].copy(), otgt) for c in (range(C) if C < 100 else (0, 1, C // 2, C - 1))}
        for _ in range(n):
            info, _ = kern(state, eps, immg, *extra)
            state = info.state._replace(momentum=None)
            for c in (range(C) if C < 100 else (0, 1, C // 2, C - 1)):  # (the oracle is a Python loop: a few chains of a big call)
                o = okern[c](ostate[c], eps, imm, *extra)
                ostate[c] = o.state._replace(momentum=None)
                ctx = dict(case=case, sampler=sampler, form=form, metric=mk, D=D, C=C, c=c, **opts)
                np.testing.assert_allclose(info.state.position[c].cpu().numpy(), o.state.position, rtol=RTOL, atol=1e-11, err_msg=str(ctx))
                np.testing.assert_allclose(info.state.potential_energy[c].item(), o.state.potential_energy, rtol=RTOL, atol=1e-11, err_msg=str(ctx))
                np.testing.assert_allclose(info.state.potential_energy_grad[c].cpu().numpy(), o.state.potential_energy_grad, rtol=RTOL,
                                           atol=1e-10, err_msg=str(ctx))
                np.testing.assert_allclose(info.acceptance_probability[c].item(), o.acceptance_probability, rtol=1e-7, atol=1e-12, err_msg=str(ctx))
                assert bool(info.is_diverging[c]) == bool(o.is_diverging), ctx
                if sampler == "nuts":
                    assert info.n_leapfrog[c].item() == o.n_leapfrog and info.num_doublings[c].item() == o.num_doublings, ctx
                    assert bool(info.is_turning[c]) == bool(o.is_turning), ctx
    finally:
        eng.set_option("resident_nuts", 2)
        eng.set_option("fused_hmc", 1)
        eng.set_option("block_dense", 1)


t0, n, bad = time.time(), 0, []


def cases():
    """FUZZ_CASES=id,id,... : exactly those; FUZZ_COUNT=N : the N ids from seed * 10**6 on (both deterministic -- what the test
    suite runs); otherwise ids from seed * 10**6 on for `seconds` of wall clock (the long sweeps recorded under profiles/)."""
    only = [int(x) for x in os.environ.get("FUZZ_CASES", "").split(",") if x]
    if only:
        yield from only
    elif os.environ.get("FUZZ_COUNT"):
        yield from range(seed0 * 1_000_000, seed0 * 1_000_000 + int(os.environ["FUZZ_COUNT"]))
    else:
        case = seed0 * 1_000_000
        while time.time() - t0 < budget:
            yield case
            case += 1


for case in cases():
    try:
        one(case)
        n += 1
    except Exception as e:
        bad.append(case)
        print("MISMATCH case", case, repr(e)[:700], flush=True)
        traceback.print_exc(limit=1)
print(f"fuzz_custom: {n} configurations in {time.time() - t0:.0f} s, {len(bad)} mismatches: {bad}")
sys.exit(1 if bad else 0)
