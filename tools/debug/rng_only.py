"""k_rng_normals alone (4096 chains x 10000 normals per call), for rocprofv3 counters."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream
from aehmc_amd.engine import get_engine
C = 4096
eng = get_engine()
rng = torch.as_tensor(RandomStream(seeds=list(range(C))).sites(1).astype(np.int64).reshape(C, 4), device="cuda").contiguous()
for _ in range(5):
    eng.rng_normals(rng, 10000)
torch.cuda.synchronize()
