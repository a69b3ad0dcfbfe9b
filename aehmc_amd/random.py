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
    """``RandomStream(seed)`` (one chain) or ``RandomStream(seeds=[...])`` (one seed per chain).

    Resuming: a kernel returns ``updates = {srng: states}`` with the generator states ``[C, n, 4]`` of
    its call sites after the transition (the reference threads them through ``updates`` the same way,
    README.md:49-51, nuts.py:138-153).  ``RandomStream.from_state(states, seeds=...)`` rebuilds a stream
    whose NEXT kernel continues exactly there -- save ``states.cpu().numpy()`` with the chain state,
    restore later, get the transitions an unbroken run would have produced."""

    def __init__(self, seed=None, seeds=None):
        if seeds is None:
            seeds = [seed]
            self.batched = False
        else:
            self.batched = True
        self.seeds = [int(s) if s is not None else None for s in np.atleast_1d(seeds).tolist()]
        self._seed_seqs = [np.random.SeedSequence(s) for s in self.seeds]
        self._n_spawned = 0
        self._resume = None  # generator states handed to the next kernel instead of fresh spawns

    @classmethod
    def from_state(cls, states, seed=None, seeds=None, batched=None, n_spawned=0):
        """Stream whose next ``sites(n)`` returns the saved states ``[C, n, 4]`` (uint64 / int64 bit
        patterns; a device tensor, numpy array or nested list of Python ints) instead of spawning.  With the
        original ``seed`` / ``seeds`` the spawn counter moves on as well, so kernels built AFTER the resumed one
        get the same call sites as in an unbroken session -- when the resumed kernel was not the first one built
        from the stream, pass ``n_spawned`` = the number of call sites handed out BEFORE it (``stream.n_spawned``
        of the original session at that point: 4 per NUTS kernel, 2 per HMC kernel); without the seeds a further
        kernel raises."""
        if hasattr(states, "detach"):
            states = states.detach().cpu().numpy()
        arr = states if isinstance(states, np.ndarray) else np.array(states, dtype=object)  # (lists: exact Python ints)
        if arr.dtype == object:  # nested lists mixing negative values and values >= 2**63: no silent truncation
            if not all(isinstance(v, (int, np.integer)) for v in arr.reshape(-1)):
                raise ValueError("generator states must be integers (64-bit patterns)")
            flat = [int(v) for v in arr.reshape(-1)]
            if any(v < -(1 << 63) or v > _M64 for v in flat):
                raise ValueError("generator states must be 64-bit patterns (int64 or uint64)")
            arr = np.array([v & _M64 for v in flat], dtype=np.uint64).reshape(arr.shape)
        if arr.dtype.kind not in "iu" or arr.dtype.itemsize != 8:
            raise ValueError(f"generator states must be int64 or uint64 bit patterns, got dtype {arr.dtype}")
        arr = np.ascontiguousarray(arr)
        if arr.dtype != np.uint64:
            arr = arr.view(np.uint64)
        if arr.ndim != 3 or arr.shape[2] != 4:
            raise ValueError(f"generator states must be [C, n_sites, 4], got {arr.shape}")
        C = arr.shape[0]
        if seeds is None and seed is None:
            self = cls(seeds=[None] * C)
            self._seed_seqs = None
            self.batched = (C > 1) if batched is None else bool(batched)
        else:
            self = cls(seed=seed, seeds=seeds)
            if self.num_chains != C:
                raise ValueError(f"{self.num_chains} seeds for {C} saved chains")
            if batched is not None:
                self.batched = bool(batched)
            if n_spawned:  # the call sites the original session had handed out before the resumed kernel
                for ss in self._seed_seqs:
                    ss.spawn(int(n_spawned))
                self._n_spawned = int(n_spawned)
        self._resume = arr.copy()
        return self

    @property
    def n_spawned(self) -> int:
        """Call sites handed out so far (save it with the states when the kernel is not the stream's first)."""
        return self._n_spawned

    @property
    def num_chains(self) -> int:
        return len(self.seeds)

    def sites(self, n: int) -> np.ndarray:
        """Next ``n`` RNG call sites for every chain -> uint64 [C, n, 4]."""
        if self._resume is not None:
            if self._resume.shape[1] != n:
                raise ValueError(f"the saved state holds {self._resume.shape[1]} call sites, the kernel being built "
                                 f"has {n} (NUTS: 4, HMC: 2)")
            out, self._resume = self._resume, None
            if self._seed_seqs is not None:  # keep the spawn sequence where the unbroken session would be
                for ss in self._seed_seqs:
                    ss.spawn(n)
                self._n_spawned += n
            return out
        if self._seed_seqs is None:
            raise ValueError("this stream was rebuilt from saved states without its seeds: it can resume ONE kernel; "
                             "pass seed= / seeds= to RandomStream.from_state to build further ones")
        out = np.empty((self.num_chains, n, 4), dtype=np.uint64)
        for c, ss in enumerate(self._seed_seqs):
            for k, child in enumerate(ss.spawn(n)):
                st = np.random.PCG64(child).state["state"]
                out[c, k] = (st["state"] >> 64, st["state"] & _M64, st["inc"] >> 64, st["inc"] & _M64)
        self._n_spawned += n
        return out
