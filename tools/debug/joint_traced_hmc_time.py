#!/usr/bin/env python3
"""HMC throughput of Neal's funnel written as a PYTHON logprob_fn above 64 coordinates: k_hmc_fused compiled against the
traced program (chain in registers, program rows in LDS; engine option joint_resident = 1, the default) beside
k_hmc_joint_rows (joint_resident = 0: the chain's rows in L2).  usage: joint_traced_hmc_time.py [C] [L]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, hmc
from aehmc_amd.engine import get_engine
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32


def funnel(q):
    v, x = q[0], q[1:]
    return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()


eng = get_engine()
for D in (100, 256, 512, 1000):
    for mode in (1, 0):
        eng.set_option("joint_resident", mode)
        r = np.random.default_rng(D)
        q0 = torch.as_tensor(0.3 * r.standard_normal((C, D)), device="cuda")
        imm = torch.ones(D, dtype=torch.float64, device="cuda")
        kernel = hmc.new_kernel(RandomStream(seeds=list(range(C))), funnel)
        state = hmc.new_state(q0, funnel)
        state = kernel.sample(state, 0.02, imm, L, 2, keep_samples=False)[1].state._replace(momentum=None)
        T = 10
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, 0.02, imm, L, T, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"funnel D={D} C={C} L={L} joint_resident={mode}: {dt / T * 1e3:.2f} ms/transition, {C * L * T / dt:.3e} leapfrog/s, "
              f"mean acceptance {float(acc.mean()):.3f}", flush=True)
eng.set_option("joint_resident", 1)
