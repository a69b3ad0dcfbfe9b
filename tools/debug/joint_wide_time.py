#!/usr/bin/env python3
"""NUTS throughput of a JOINT user-defined density (Neal's funnel, density only) below and above 64 coordinates: the
single-launch kernel (D <= 64: one coordinate per lane) against the lock-step path (row in LDS, ceil(D / 64) forward
passes per gradient).  usage: joint_wide_time.py [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SRC = """
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  auto v = q[0];
  auto lp = -v * v / 18.0;
  for (int i = 1; i < q.size(); i++) lp += -0.5 * q[i] * q[i] * exp(-v) - 0.5 * v;
  return lp;
}
"""
for D in (64, 10, 10, 32, 64, 100, 256, 1000):
    r = np.random.default_rng(D)
    tgt = targets.CustomJoint(SRC, dim=D)
    q0 = torch.as_tensor(0.3 * r.standard_normal((C, D)), device="cuda")
    imm = torch.ones(D, dtype=torch.float64, device="cuda")
    kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
    state = nuts.new_state(q0, tgt)
    eps = 0.05
    for _ in range(2):
        state = kernel(state, eps, imm)[0].state._replace(momentum=None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); nl = 0
    T = 5
    for _ in range(T):
        info = kernel(state, eps, imm)[0]
        state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"funnel D={D} C={C}: {dt / T * 1e3:.2f} ms/transition, {nl / T / C:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)
