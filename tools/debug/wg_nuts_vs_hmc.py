#!/usr/bin/env python3
"""Workgroup-per-chain GLM kernels: cost of a leapfrog under NUTS against HMC when every chain does the SAME number of
leapfrogs (tiny step size, trees cut at `max_num_expansions`: 2^k - 1 leapfrogs each, no U-turn) -- what is left of the gap
to HMC in a real run is the spread of tree sizes.  usage: wg_nuts_vs_hmc.py [N] [D] [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
GLM = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
rng = np.random.default_rng(0)
X = rng.normal(size=(N, D)); w = rng.normal(size=D) / np.sqrt(D)
y = (rng.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
tgt = targets.CustomGLM(GLM, torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda"))
q0 = torch.as_tensor(w + 0.1 * rng.standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
eps = 1e-3 / np.sqrt(N)
for k in (3, 4, 5):
    L = 2 ** k - 1
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=k)
    state = nuts.new_state(q0, tgt)
    state = kernel.sample(state, eps, imm, 1, keep_samples=False)[1].state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, info, acc, div = kernel.sample(state, eps, imm, 4, keep_samples=False)
    torch.cuda.synchronize(); dtn = time.perf_counter() - t0
    nl = int(info.n_leapfrog.sum())
    hk = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    hs = hmc.new_state(q0, tgt)
    hs = hk.sample(hs, eps, imm, L, 1, keep_samples=False)[1].state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hk.sample(hs, eps, imm, L, 4, keep_samples=False)
    torch.cuda.synchronize(); dth = time.perf_counter() - t0
    print(f"N={N} D={D} C={C} depth {k}: NUTS {nl / 4 / C:.1f} leapfrogs/chain, {dtn / nl * C * 1e3:.3f} ms per leapfrog of all chains; "
          f"HMC L={L}: {dth / (4 * L) * 1e3:.3f} ms per leapfrog of all chains", flush=True)
