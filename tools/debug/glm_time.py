#!/usr/bin/env python3
"""NUTS throughput of a user-defined row-reduction target (logistic regression, density only) on the one-launch kernels
(D <= 32: the wavefront that owns a chain sweeps the data rows itself) against the lock-step path (two chain-batched
GEMMs with the data matrix per leapfrog).  usage: glm_time.py [N] [D] [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
from aehmc_amd.engine import get_engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
SRC = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
if os.environ.get("LOSS") == "quadratic":  # the same sweep with a row function of three instructions: what the data traffic alone costs
    SRC = SRC.replace("return y * z - softplus(z);", "return -0.5 * (y - z) * (y - z);")
if os.environ.get("LOSS") == "hand":  # value and derivative from ONE exponential, written by hand
    SRC = """
__device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &l, double &d) {
  const double e = exp(-fabs(z)), s = 1.0 / (1.0 + e);
  l = (z > 0 ? z : 0.0) + log1p(e) - y * z;
  d = (z > 0 ? s : 1.0 - s) - y;
}
__device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g) { u = 0.5 * q * q / 4.0; g = q / 4.0; }
"""
r = np.random.default_rng(0)
X = r.normal(size=(N, D)); w = r.normal(size=D)
y = (r.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
eng = get_engine()
from aehmc_amd import hmc
for resident in ((2, 2, 0) if not os.environ.get('LOSS') else (2, 2)):
    eng.set_option("resident_nuts", resident)
    eng.set_option("fused_hmc", 1 if resident else 0)
    tgt = targets.CustomGLM(SRC, torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda"))
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    state = nuts.new_state(torch.as_tensor(w + 0.1 * r.standard_normal((C, D)), device="cuda"), tgt)
    eps, imm = 0.3 / np.sqrt(N), torch.ones(D, dtype=torch.float64, device="cuda")
    for _ in range(2):
        state = kernel(state, eps, imm)[0].state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nl = 0
    T = 4
    for _ in range(T):
        info = kernel(state, eps, imm)[0]
        state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"logistic N={N} D={D} C={C} resident_nuts={resident}: NUTS one transition per call {dt / T * 1e3:.2f} ms/transition, {nl / T / C:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = kernel.sample(state, eps, imm, 12)
    nl = int(out[1].n_leapfrog.sum()) if out[1].n_leapfrog.numel() else 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"    NUTS sample(12): {dt / 12 * 1e3:.2f} ms/transition", flush=True)
    hk = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    hs = hmc.new_state(state.position, tgt)
    hk.sample(hs, eps, imm, 16, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hk.sample(hs, eps, imm, 16, 6)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"    HMC L=16 sample(6): {dt / 6 * 1e3:.2f} ms/transition, {C * 16 * 6 / dt:.3e} leapfrog/s", flush=True)
eng.set_option("resident_nuts", 2)
eng.set_option("fused_hmc", 1)
