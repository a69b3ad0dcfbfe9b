"""RandomStream look-alike reproducing the reference's RNG ("scheme A").

aesara's ``RandomStream(seed)`` keeps ``SeedSequence(seed)``; every ``srng.<dist>()`` call
site, in graph-construction order, owns ``default_rng(seedseq.spawn(1)[0])`` (PCG64) and
advances it each time the compiled graph runs.  Here a stream holds one SeedSequence per
chain; ``sites(n)`` hands the next ``n`` call sites to a kernel as PCG64 states
``[C, n, 4]`` (state_hi, state_lo, inc_hi, inc_lo) that the HIP kernels advance in place
exactly as numpy's Generator would.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


class RandomStream:
    def __init__(self, seed=None, seeds=None):
        if seeds is None:
            seeds = [seed]
            self.batched = False
        else:
            self.batched = True
        self.seeds = [int(s) if s is not None else None for s in np.atleast_1d(seeds).tolist()]
        self._seed_seqs = [np.random.SeedSequence(s) for s in self.seeds]
        self._n_spawned = 0

    @property
    def num_chains(self) -> int:
        return len(self._seed_seqs)

    def sites(self, n: int) -> np.ndarray:
        """Next ``n`` RNG call sites for every chain -> uint64 [C, n, 4]."""
        out = np.empty((self.num_chains, n, 4), dtype=np.uint64)
        for c, ss in enumerate(self._seed_seqs):
            for k, child in enumerate(ss.spawn(n)):
                st = np.random.PCG64(child).state["state"]
                out[c, k] = (st["state"] >> 64, st["state"] & _M64, st["inc"] >> 64, st["inc"] & _M64)
        self._n_spawned += n
        return out
