#!/usr/bin/env python3
"""Compile (hipRTC) + bind time of a joint user-defined target with an empty code-object cache: the base program (new_state,
the lock-step density kernel and the two one-launch kernels above 64 coordinates) and the first NUTS step (the
single-launch kernel below 64).  Measured: 1.4 s + 1.6 s."""
import os, sys, time, tempfile
os.environ["AEHMC_AMD_RTC_CACHE"] = tempfile.mkdtemp()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from aehmc_amd import RandomStream, nuts, targets
from test_gpu_autodiff import FUNNEL
torch.zeros(1, device="cuda")
for D in (10, 100):
    tgt = targets.CustomJoint(FUNNEL + f"// {D}\n", dim=D)
    q0 = torch.as_tensor(0.3 * np.random.default_rng(0).normal(size=(8, D)), device="cuda")
    t0 = time.perf_counter(); st = nuts.new_state(q0, tgt); torch.cuda.synchronize(); t1 = time.perf_counter()
    k = nuts.new_kernel(RandomStream(seeds=list(range(8))), tgt, max_num_expansions=4)
    k(st, 0.1, np.ones(D)); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"D={D}: bind + new_state {t1 - t0:.2f} s, first NUTS step {t2 - t1:.2f} s")
