#!/usr/bin/env python3
"""The workgroup-per-chain kernels compiled for 3 / 4 wavefronts per SIMD (engine option wg_waves): logistic regression by
density only (targets.CustomGLM) and as a Python function (traced, k_nuts_joint_wg), N = 1e5 rows, 1024 chains.
usage: wg_waves_time.py [D ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc, nuts, targets, tracing
from aehmc_amd.engine import get_engine
eng = get_engine()
N, C = 100_000, 1024
GLM = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
for D in [int(x) for x in sys.argv[1:]] or [8, 16, 32]:
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)); w = rng.normal(size=D) / np.sqrt(D)
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)

    def logistic(q):
        z = X @ q
        return (y * z - tracing.softplus(z)).sum() - 0.5 * (q @ q) / 4.0

    for form in ("CustomGLM", "python"):
        for waves in (3, 4, 0):
            eng.set_option("wg_waves", waves)
            tgt = (targets.CustomGLM(GLM, torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")) if form == "CustomGLM"
                   else targets.from_callable(logistic, D))
            q0 = torch.as_tensor(w + 0.1 * rng.standard_normal((C, D)), device="cuda")
            imm = torch.ones(D, dtype=torch.float64, device="cuda")
            eps = 0.3 / np.sqrt(N)
            out = []
            kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
            state = nuts.new_state(q0, tgt)
            state = kernel.sample(state, eps, imm, 2, keep_samples=False)[1].state._replace(momentum=None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            samples, info, acc, div = kernel.sample(state, eps, imm, 4, keep_samples=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            nl = int(info.n_leapfrog.sum())
            out.append(f"NUTS {dt / 4 * 1e3:.2f} ms/transition {nl / dt:.3e} leapfrog/s")
            hk = hmc.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            hs = hmc.new_state(q0, tgt)
            hs = hk.sample(hs, eps, imm, 16, 1, keep_samples=False)[1].state._replace(momentum=None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            hk.sample(hs, eps, imm, 16, 3, keep_samples=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.append(f"HMC L=16 {dt / 3 * 1e3:.2f} ms/transition {C * 16 * 3 / dt:.3e} leapfrog/s")
            print(f"logistic N={N} D={D} C={C} {form:9s} wg_waves={waves}: " + "; ".join(out), flush=True)
eng.set_option("wg_waves", 0)
