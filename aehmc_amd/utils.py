"""``RaveledParamsMap`` (reference: aehmc/utils.py:22-74): maps a set of named parameters to the vector of their raveled
values and back -- how a model with several parameter blocks is handed to ``hmc`` / ``nuts``, which work on ONE position
vector.  The reference does this on symbolic variables (shapes inferred from the graph); here the parameters are eager
arrays, so the map is built from reference VALUES (anything with ``shape`` / ``dtype``: numpy arrays, torch tensors,
Python scalars) keyed by name.  ``batch_ndim=1`` treats a leading axis as the chain axis ([C, ...] <-> [C, total])."""
from __future__ import annotations

from typing import Dict, Iterable, Mapping

import numpy as np


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


class RaveledParamsMap:
    """Maps a set of named arrays to a vector of their raveled values (aehmc/utils.py:22-74)."""

    def __init__(self, ref_params, batch_ndim: int = 0):
        if isinstance(ref_params, Mapping):
            items = list(ref_params.items())
        else:  # iterable of (name, value) pairs or of objects with a ``name``
            items = [(p if isinstance(p, tuple) else (getattr(p, "name"), p)) for p in ref_params]
        self.batch_ndim = int(batch_ndim)
        self.ref_params = tuple(k for k, _ in items)
        self.ref_shapes = [tuple(np.shape(v))[self.batch_ndim:] for _, v in items]
        self.ref_dtypes = [getattr(v, "dtype", np.asarray(v).dtype) for _, v in items]
        sizes = [int(np.prod(s)) if len(s) else 1 for s in self.ref_shapes]
        ends = np.cumsum(sizes).tolist()
        self.slice_indices = list(zip([0] + ends[:-1], ends))
        self.vec_slices = [slice(*idx) for idx in self.slice_indices]
        self.size = ends[-1] if ends else 0

    def ravel_params(self, params: Iterable):
        """Concatenate the raveled vectors of each parameter (aehmc/utils.py:55-57), in the map's order; ``params`` is a
        sequence in that order or a mapping by name."""
        if isinstance(params, Mapping):
            params = [params[k] for k in self.ref_params]
        params = list(params)
        b = self.batch_ndim
        if any(_is_torch(p) for p in params):
            import torch
            ts = [p if _is_torch(p) else torch.as_tensor(np.asarray(p)) for p in params]
            dt = torch.result_type(ts[0], ts[0])
            for t in ts[1:]:
                dt = torch.promote_types(dt, t.dtype)
            dev = next(t.device for t in ts if t.is_cuda) if any(t.is_cuda for t in ts) else ts[0].device
            return torch.cat([t.to(device=dev, dtype=dt).reshape(tuple(t.shape[:b]) + (-1,)) for t in ts], dim=-1)
        arrs = [np.asarray(p) for p in params]
        return np.concatenate([a.reshape(a.shape[:b] + (-1,)) for a in arrs], axis=-1)

    def unravel_params(self, raveled_params) -> Dict:
        """Unravel a concatenated set of raveled parameters into ``{name: array}`` with the reference shapes and dtypes
        (aehmc/utils.py:59-71)."""
        lead = tuple(raveled_params.shape[:-1])
        if raveled_params.shape[-1] != self.size:
            raise ValueError(f"expected a vector of {self.size} raveled values, got {raveled_params.shape[-1]}")
        out = {}
        for k, slc, s, t in zip(self.ref_params, self.vec_slices, self.ref_shapes, self.ref_dtypes):
            v = raveled_params[..., slc].reshape(lead + s)
            if _is_torch(v):
                import torch
                out[k] = v.to(t if isinstance(t, torch.dtype) else getattr(torch, np.dtype(t).name))
            else:
                out[k] = v.astype(t if not _is_torch_dtype(t) else str(t).split(".")[-1])
        return out

    def __repr__(self):
        return f"{type(self).__name__}(({', '.join(str(k) for k in self.ref_params)}))"


def _is_torch_dtype(t):
    return type(t).__module__.split(".")[0] == "torch"
