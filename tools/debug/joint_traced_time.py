#!/usr/bin/env python3
"""NUTS throughput of Neal's funnel written as a PYTHON logprob_fn (traced: aehmc_amd/tracing.py) -- above 64 coordinates its
reverse-mode program, one sweep per gradient -- beside the same density as a hand-written HIP template (forward mode:
ceil(D / 64) passes).  usage: joint_traced_time.py [C]"""
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


def funnel(q):
    v, x = q[0], q[1:]
    return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()


for D in (10, 64, 100, 256, 1000):
    for form in ("python", "template"):
        if form == "template" and D > 1000:
            continue
        r = np.random.default_rng(D)
        tgt = targets.from_callable(funnel, D) if form == "python" else targets.CustomJoint(SRC, dim=D)
        q0 = torch.as_tensor(0.3 * r.standard_normal((C, D)), device="cuda")
        imm = torch.ones(D, dtype=torch.float64, device="cuda")
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        eps = 0.05
        for _ in range(2):
            state = kernel(state, eps, imm)[0].state._replace(momentum=None)
        T = 5
        state = kernel.sample(state, eps, imm, 2, keep_samples=False)[1].state._replace(momentum=None)  # (compiles the sample() program)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, eps, imm, T, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum())
        print(f"funnel D={D} C={C} {form:8s}: {dt / T * 1e3:.2f} ms/transition, {nl / T / C:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)
