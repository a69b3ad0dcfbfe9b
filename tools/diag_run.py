#!/usr/bin/env python3
"""Fixed workload for rocprofv3: the secondary lines of bench.py (diagonal-mass NUTS and HMC at
D = 1e4, 4096 chains), a few transitions each.
usage: python3 tools/diag_run.py [nuts|hmc|both] [transitions] [D] [C] [fp_contract]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import diag_case

what = sys.argv[1] if len(sys.argv) > 1 else "both"
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # engine calls (HMC: bench.HMC_PER_CALL transitions each)
D = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
if len(sys.argv) > 5 and int(sys.argv[5]):
    from aehmc_amd.engine import get_engine
    get_engine().set_option("fp_contract", 1)
for kind in ("nuts", "hmc"):
    if what not in (kind, "both"):
        continue
    state, step = diag_case(kind, D, C, torch.device("cuda"))
    nl = 0
    for _ in range(NT):
        info, _ = step(state)
        state = info.state._replace(momentum=None)
        nl += int(info.n_leapfrog.sum().item())
    torch.cuda.synchronize()
    print(f"{kind} D={D} C={C}: {nl} leapfrogs in {NT} transitions")
