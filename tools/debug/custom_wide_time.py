#!/usr/bin/env python3
"""NUTS throughput of a user-defined coordinate-wise target (Student-t, density only: differentiated by the engine) on the
workgroup-per-chain kernel against the built-in diagonal Gaussian at the same shape.  usage: custom_wide_time.py [D] [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
D = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
SRC = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const double nu = prm[0][i], s = prm[1][i];
  const T z = q / s;
  return -0.5 * (nu + 1.0) * log1p(z * z / nu);
}
"""
# the same density with the divisions by parameters done once on the host (1 / s, 1 / nu as parameters): a double
# division is ~25 instructions on this GPU and forward mode doubles each one
SRC_MUL = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const double hn = prm[0][i], inv_s = prm[1][i], inv_nu = prm[2][i];
  const T z = q * inv_s;
  return hn * log1p(z * z * inv_nu);
}
"""
# a density of the built-in Gaussian's own cost (one division by sigma in value and derivative each): what the custom PATH costs
SRC_GAUSS = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const T z = q / prm[0][i];
  return -0.5 * (z * z) - prm[1][i];
}
"""
r = np.random.default_rng(0)
nu, s = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
q0 = torch.as_tensor(r.standard_normal((C, D)), device="cuda")
imm = torch.ones(D, dtype=torch.float64, device="cuda")
for name, tgt in (("custom Student-t (density only)", targets.Custom(SRC, params=[nu, s])),
                  ("custom Student-t, reciprocals as parameters", targets.Custom(SRC_MUL, params=[-0.5 * (nu + 1.0), 1.0 / s, 1.0 / nu])),
                  ("custom Gaussian (density only)", targets.Custom(SRC_GAUSS, params=[s, np.log(s) + 0.9189385332046727])),
                  ("built-in diagonal Gaussian", targets.DiagGaussian(np.zeros(D), s))):
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = nuts.new_state(q0, tgt)
    eps = 0.4 * D ** -0.25
    for _ in range(2):
        info, _ = kernel(state, eps, imm)
        state = info.state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nl = 0
    for _ in range(5):
        info, _ = kernel(state, eps, imm)
        state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"D={D} C={C} {name}: {dt / 5 * 1e3:.2f} ms/transition, {nl / dt:.3e} leapfrog/s", flush=True)
