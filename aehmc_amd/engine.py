"""Host-side engine object: owns one C-ABI ctx on one GPU and the torch-allocated device
buffers handed to it.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import collections
import ctypes as ct
import hashlib

import numpy as np
import torch

try:  # fast content hash for large host-side matrices (offline wheelhouse); hashlib otherwise
    import xxhash
except ImportError:  # pragma: no cover
    xxhash = None

from . import _lib
from .targets import Target


# Host arrays are keyed by their FULL content at any size (the reference re-reads the matrix on every call, so an
# in-place edit must be seen).  `SAMPLED_HASH_ABOVE` (bytes; None = never) is an explicit opt-in for callers who
# promise not to edit large arrays in place: above it only a strided sample is hashed.
SAMPLED_HASH_ABOVE = None
_WARN_HOST_BYTES = 64 << 20
_warned_big_host_array = False


def _digest(buf):
    if xxhash is not None:
        return xxhash.xxh3_128_hexdigest(buf)
    return hashlib.blake2b(buf, digest_size=16).hexdigest()


def _content_key(arr: np.ndarray):
    """Key a host array by CONTENT: numpy inputs may be edited in place between calls (the
    reference re-reads the matrix on every call), so neither id() nor a sample of the elements may
    hit the cache.  The whole array is hashed (xxh3: ~10 GB/s, i.e. ~0.1 s for the 800 MB matrix of
    D = 1e4 -- once per step() call; a warning says once that a torch tensor on the device, keyed by
    identity and version counter, avoids both the hashing and the upload).  With the module-level
    opt-in ``SAMPLED_HASH_ABOVE`` set, larger arrays are keyed by shape plus the hashes of a 1/256
    strided sample and of the first and last MB (then an in-place edit that misses the sample needs
    ``force=True``)."""
    global _warned_big_host_array
    a = np.ascontiguousarray(arr)
    buf = memoryview(a).cast("B")
    if a.nbytes > _WARN_HOST_BYTES and not _warned_big_host_array:
        _warned_big_host_array = True
        import warnings
        warnings.warn(f"aehmc_amd: a {a.nbytes >> 20} MB host array is hashed on every call and re-uploaded when it "
                      "changes; pass a torch tensor on the device to skip both", stacklevel=4)
    if SAMPLED_HASH_ABOVE is None or a.nbytes <= SAMPLED_HASH_ABOVE:
        return _digest(buf)
    flat = a.reshape(-1)
    sample = np.ascontiguousarray(flat[::256])
    mb = (1 << 20) // a.itemsize
    return ("sampled", a.shape, _digest(memoryview(sample).cast("B")), _digest(memoryview(flat[:mb]).cast("B")),
            _digest(memoryview(flat[-mb:]).cast("B")))


def _param_key(v):
    """Cache key of one target / metric parameter: DEVICE tensors by identity + version counter, anything
    else (CPU tensors, numpy arrays, lists, scalars) by content.  (A CPU tensor can be edited through a numpy view of
    its storage without its version counter moving -- `t.numpy()[0] = 1.0` -- so identity + version is not a safe key
    for it; it is uploaded anyway, hashing it is the cheaper part.)"""
    if isinstance(v, torch.Tensor):
        if v.device.type == "cpu":
            arr = v.detach().to(torch.float64).numpy()
            return ("c", arr.shape, _content_key(arr))
        return ("t", id(v), v._version, tuple(v.shape))
    arr = np.asarray(v, dtype=np.float64)
    return ("n", arr.shape, _content_key(arr))


def _asymmetry(t: torch.Tensor) -> float:
    """max |t - t^T| in row blocks (no D x D temporary)."""
    worst, D = 0.0, t.shape[0]
    for lo in range(0, D, 1024):
        hi = min(D, lo + 1024)
        worst = max(worst, float((t[lo:hi] - t[:, lo:hi].T).abs().max()))
    return worst


def _dev_f64(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), device=device)


class EngineError(RuntimeError):
    pass


class PerChain:
    """Marks a step size [C] or a diagonal inverse mass matrix [C] / [C, D] as holding one
    value (row) per chain -- what per-chain window adaptation produces.  The reference has
    no chain axis, so a plain [C, D] array would be ambiguous with a dense [D, D] matrix."""

    def __init__(self, value, sqrt_mass=None):
        self.value = value
        self.sqrt_mass = sqrt_mass


class Engine:
    """One aehmc_ctx.  Not thread-safe (as the C-ABI)."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise EngineError("aehmc_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.lib = _lib.load()
        ctx = ct.c_void_p()
        rc = self.lib.aehmc_create(ct.byref(ctx), self.device.index or 0)
        self.ctx = ctx
        if rc:
            raise EngineError(f"aehmc_create failed: {self._err()}")
        self._target_key = None
        self._metric_key = None
        self.rtc_cache_dir = self._rtc_cache_dir()
        if self.rtc_cache_dir:
            self._check(self.lib.aehmc_set_rtc_cache(self.ctx, self.rtc_cache_dir.encode()), "aehmc_set_rtc_cache")
        # metric handles: (device imm, device sqrt-mass) per metric content, so that kernels which alternate between
        # metrics on one device (each kernel its own dense mass matrix) factor every matrix ONCE
        self._metric_cache = collections.OrderedDict()
        self.metric_cache_bytes = 4 << 30
        self.n_metric_factorizations = 0
        self._keep = {}
        self._ws = None
        self.D = None
        self.metric_ndim = None
        self.metric_D = None

    @staticmethod
    def _rtc_cache_dir():
        """Where the compiled code objects of user-defined targets are kept across processes: AEHMC_AMD_RTC_CACHE (a
        directory; "0" / "" = off), default ~/.cache/aehmc_amd/rtc-<hash of the library's sources> -- the headers a
        run-time compiled program includes are part of what it was compiled from."""
        import os
        from . import _build
        where = os.environ.get("AEHMC_AMD_RTC_CACHE")
        if where is not None and where in ("", "0"):
            return None
        try:
            base = where or os.path.join(os.path.expanduser("~"), ".cache", "aehmc_amd")
            path = os.path.join(base, "rtc-" + _build.source_hash()[:16])
            os.makedirs(path, exist_ok=True)
            return path
        except OSError:
            return None

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.aehmc_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def _err(self):
        return self.lib.aehmc_last_error(self.ctx).decode()

    def _check(self, rc, what):
        if rc:
            raise EngineError(f"{what} failed ({rc}): {self._err()}")

    @property
    def stream(self):
        return ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ binding
    def set_target(self, target, D: int, force: bool = False, scalar=None):
        # a Python function of the position (the reference's logprob_fn): traced once into a Custom / CustomJoint target
        if not isinstance(target, Target):
            from . import targets as _targets
            target = _targets.as_target(target, D, scalar)
        # keyed by the CONTENT of the parameters (numpy arrays edited in place between calls must be seen,
        # as for the metric below), not by the identity of the Target object
        params = target.params()
        key = (type(target), D, tuple((k, _param_key(v)) for k, v in sorted(params.items())))
        if getattr(target, "source", None) is not None:
            key = key + (target.source,)
        if self._target_key == key and not force:
            return
        if target.dim is not None and target.dim != D:
            raise ValueError(f"target has dimension {target.dim}, position has {D}")
        p = {k: _dev_f64(v, self.device) for k, v in params.items()}
        if getattr(target, "source", None) is not None:  # user-defined coordinate-wise target: compiled with hipRTC
            import os
            arrs = [p[f"p{k}"] for k in range(len(target.param_list))]
            ptrs = (ct.c_void_p * max(len(arrs), 1))(*[a.data_ptr() for a in arrs])
            inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
            # a failed compile leaves the ctx bound to what it was bound to (aehmc_set_custom_target is a
            # transaction): the arrays of THAT binding must stay alive, so `_keep` changes only on success
            if "X" in p:  # row-reduction target over a data matrix
                X, y = p["X"].contiguous(), p["y"].reshape(-1).contiguous()
                if X.ndim != 2 or X.shape[1] != D or y.numel() != X.shape[0]:
                    raise ValueError(f"GLM target: X must be [N, {D}] and y [N], got {tuple(X.shape)} and {tuple(y.shape)}")
                keep = (target, p, X, y)
                self._check(self.lib.aehmc_set_custom_glm_target(self.ctx, target.source.encode(), D, X.shape[0],
                                                                 X.data_ptr(), y.data_ptr(), ptrs, len(arrs), inc.encode()),
                            "aehmc_set_custom_glm_target")
            elif target.kind == 7:  # targets.T_JOINT: a joint density, differentiated by the engine (D <= 2048; single-launch kernels up to 64)
                keep = (target, p)
                self._check(self.lib.aehmc_set_custom_joint_target(self.ctx, target.source.encode(), D, ptrs, len(arrs),
                                                                   inc.encode()), "aehmc_set_custom_joint_target")
            else:
                keep = (target, p)
                self._check(self.lib.aehmc_set_custom_target(self.ctx, target.source.encode(), D, ptrs, len(arrs),
                                                             inc.encode()), "aehmc_set_custom_target")
            self._keep["target"] = keep
            self._target_key, self.D = key, D
            self._ws = None
            return
        c = _lib.CTarget(kind=target.kind, D=D, N=0)
        for name in ("mu", "sigma", "prec", "X", "y"):
            if name in p:
                setattr(c, name, p[name].data_ptr())
        if "X" in p:
            c.N = p["X"].numel()
        self._check(self.lib.aehmc_set_target(self.ctx, ct.byref(c)), "aehmc_set_target")
        self._keep["target"] = (target, p)
        self._target_key, self.D = key, D
        self._ws = None

    def set_metric(self, inverse_mass_matrix, D: int, force: bool = False):
        """gaussian_metric(inverse_mass_matrix) -- aehmc/metrics.py:44-63."""
        imm = inverse_mass_matrix
        if isinstance(imm, PerChain):
            return self._set_metric_per_chain(imm, D)
        ndim = imm.ndim if hasattr(imm, "ndim") else np.ndim(imm)
        if ndim > 2:
            raise ValueError(
                f"Expected a mass matrix of dimension 1 (diagonal) or 2, got {ndim}")
        if isinstance(imm, torch.Tensor):
            key = (id(imm), imm._version, D)
        else:  # numpy / python scalars may be mutated in place: keyed by content, whatever the size
            arr = np.asarray(imm, dtype=np.float64)
            key = (arr.shape, _content_key(arr), D)
        if self._metric_key == key and not force:
            return
        handle = None if force else self._metric_cache.get(key)
        if handle is None:
            t = _dev_f64(imm, self.device)
            if ndim == 2:
                if t.shape != (D, D):
                    raise ValueError(f"dense inverse mass matrix must be [{D},{D}], got {tuple(t.shape)}")
                # the reference factors one triangle (metrics.py:56) and multiplies by the full matrix
                # (metrics.py:71); the two agree for a symmetric matrix.  Estimates such as A @ A.T or
                # a Welford covariance are symmetric only up to rounding: tolerate that much.
                if _asymmetry(t) > 1e-10 * float(t.diagonal().abs().max()):
                    raise ValueError("dense inverse mass matrix must be symmetric")
            else:
                t = t.reshape(-1)
                if ndim == 1 and t.numel() != D:
                    raise ValueError(f"diagonal inverse mass matrix must have {D} entries")
            # sqrt(1 / imm) / L^-T (metrics.py:45,49,56-58), once per metric content
            sm = torch.empty_like(t)
            self._check(self.lib.aehmc_metric_sqrt(self.ctx, ndim, D, t.data_ptr(), sm.data_ptr(), self.stream),
                        "aehmc_metric_sqrt")
            self.n_metric_factorizations += 1
            handle = (imm, t, sm)  # (the caller's object too: a torch tensor's id() must stay taken while cached)
            self._metric_cache[key] = handle
            held = 0
            for k in reversed(list(self._metric_cache)):  # keep the most recent handles within the byte budget
                held += 2 * self._metric_cache[k][1].numel() * 8
                if held > self.metric_cache_bytes and k != key:
                    del self._metric_cache[k]
        else:
            self._metric_cache.move_to_end(key)
        _, t, sm = handle
        c = _lib.CMetric(ndim=ndim, D=D, imm=t.data_ptr(), sqrt_mass=sm.data_ptr())
        self._keep["metric"] = handle
        self._check(self.lib.aehmc_set_metric(self.ctx, ct.byref(c)), "aehmc_set_metric")
        if self.metric_ndim != ndim:
            self._ws = None
        self._metric_key, self.metric_ndim, self.metric_D = key, ndim, D

    def _set_metric_per_chain(self, pc: PerChain, D: int):
        t = _dev_f64(pc.value, self.device)
        if t.ndim == 1:      # [C] scalar metrics (scalar positions): ndim 0
            ndim, t = 0, t.reshape(-1, 1)
        elif t.ndim == 2:    # [C, D] diagonal metrics
            ndim = 1
            if t.shape[1] != D:
                raise ValueError(f"per-chain diagonal inverse mass matrix must be [C,{D}]")
        elif t.ndim == 3:    # [C, D, D] dense metrics (is_mass_matrix_full adaptation)
            ndim = 2
            if t.shape[1:] != (D, D):
                raise ValueError(f"per-chain dense inverse mass matrix must be [C,{D},{D}]")
        else:
            raise ValueError("PerChain inverse mass matrix must be [C], [C, D] or [C, D, D]")
        if pc.sqrt_mass is not None:
            sm = _dev_f64(pc.sqrt_mass, self.device).reshape(t.shape)
        elif ndim == 2:      # L^-T per chain (metrics.py:56-58), on the device
            t = t.contiguous()
            # one wavefront factors one matrix: ~25 ms for 4096 matrices of 200 x 200.  A device tensor that has not
            # changed since the last call (identity + version counter) keeps its factors; window adaptation hands
            # sqrt_mass over itself
            key = (id(pc.value), pc.value._version, tuple(t.shape)) if isinstance(pc.value, torch.Tensor) else None
            cached = self._keep.get("pc_sqrt")
            if key is not None and cached is not None and cached[0] == key:
                sm = cached[2]
            else:
                sm = torch.empty_like(t)
                self._check(self.lib.aehmc_metric_sqrt_per_chain(self.ctx, t.shape[0], D, t.data_ptr(),
                                                                 sm.data_ptr(), self.stream),
                            "aehmc_metric_sqrt_per_chain")
                self.n_metric_factorizations += 1
                if key is not None:
                    self._keep["pc_sqrt"] = (key, pc.value, sm)  # (the tensor itself: its id() stays taken while cached)
        else:
            sm = torch.sqrt(torch.reciprocal(t))
        c = _lib.CMetric(ndim=ndim, per_chain=1, D=D, imm=t.data_ptr(), sqrt_mass=sm.data_ptr(),
                         n_chains=t.shape[0])
        self._keep["metric"] = (pc, t, sm)
        self._check(self.lib.aehmc_set_metric(self.ctx, ct.byref(c)), "aehmc_set_metric")
        if self.metric_ndim != ndim:
            self._ws = None
        self._metric_key, self.metric_ndim, self.metric_D = None, ndim, D

    def set_step_sizes(self, eps):
        """eps: PerChain -> per-chain step sizes; anything else -> scalar (returned)."""
        if isinstance(eps, PerChain):
            t = _dev_f64(eps.value, self.device).reshape(-1)
            self._keep["eps"] = t
            self._check(self.lib.aehmc_set_step_sizes(self.ctx, t.data_ptr(), t.numel()), "aehmc_set_step_sizes")
            return 0.0
        self._clear_step_sizes()
        return float(eps)

    def _clear_step_sizes(self):
        """Per-chain step sizes apply to ONE step/sample call: the ctx is shared per device and
        must not carry them into an unrelated later call."""
        self._keep.pop("eps", None)
        self._check(self.lib.aehmc_set_step_sizes(self.ctx, None, 0), "aehmc_set_step_sizes")

    def _step_call(self, fn, what, *args):
        try:
            self._check(fn(*args), what)
        finally:  # kernel arguments were captured at launch: safe to drop the binding now
            self._clear_step_sizes()

    def synchronize(self):
        """Wait for the stream and raise on device-side failures that are not data."""
        self._check(self.lib.aehmc_synchronize(self.ctx, self.stream), "aehmc_synchronize")

    def ensure_workspace(self, C: int, max_exp: int):
        need = self.lib.aehmc_workspace_bytes(self.ctx, C, max_exp)
        if need < 0:
            raise EngineError("aehmc_workspace_bytes: target/metric not set")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        self._check(self.lib.aehmc_set_workspace(self.ctx, self._ws.data_ptr(), self._ws.numel()),
                    "aehmc_set_workspace")

    def rtc_stats(self):
        """(programs compiled by hipRTC, programs loaded from the on-disk cache) of this engine's ctx."""
        import ctypes as ct
        a, b = ct.c_int64(0), ct.c_int64(0)
        self._check(self.lib.aehmc_rtc_stats(self.ctx, ct.byref(a), ct.byref(b)), "aehmc_rtc_stats")
        return a.value, b.value

    def set_option(self, name: str, value: int):
        self._check(self.lib.aehmc_set_option(self.ctx, name.encode(), int(value)), "aehmc_set_option")

    # ------------------------------------------------------------------ calls
    def new_state(self, q):
        C, D = q.shape
        U = torch.empty(C, dtype=torch.float64, device=self.device)
        g = torch.empty_like(q)
        self._check(self.lib.aehmc_new_state(self.ctx, C, q.data_ptr(), U.data_ptr(), g.data_ptr(),
                                             self.stream), "aehmc_new_state")
        return U, g

    def check_gradient(self, q, rtol=1e-5, max_coordinates=32):
        """A hand-written gradient of a user-defined target against central differences of its own potential, at the
        first chains of ``q`` (the reference differentiates ``logprob_fn`` itself, hmc.py:33-34: there a wrong gradient
        cannot exist; here it would sample the wrong distribution silently).  Raises ValueError."""
        C, D = q.shape
        qs = q[:min(C, 2)].detach()
        self.ensure_workspace(max(C, 2 * min(D, max_coordinates)), 1)  # (the perturbed positions are evaluated as chains)
        U0, g = self.new_state(qs.contiguous())
        gen = np.random.default_rng(12345)
        idx = np.arange(D) if D <= max_coordinates else np.sort(gen.choice(D, max_coordinates, replace=False))
        cols = torch.as_tensor(idx, device=self.device)
        rows = torch.arange(len(idx), device=self.device)
        worst = 0.0
        for c in range(qs.shape[0]):
            h = 1e-6 * torch.clamp(qs[c].abs(), min=1.0)[cols]
            pert = qs[c].repeat(2 * len(idx), 1)
            pert[rows, cols] += h
            pert[rows + len(idx), cols] -= h
            Up, _ = self.new_state(pert.contiguous())
            fd = (Up[:len(idx)] - Up[len(idx):]) / (2 * h)
            ga = g[c][cols]
            # (central differences: truncation of order h^2, rounding of order eps |U| / h)
            tol = rtol * torch.clamp(ga.abs(), min=1.0) + 1e-9 * U0[c].abs() / h
            err = (fd - ga).abs() / tol
            k = int(torch.argmax(err))
            worst = max(worst, float(err[k]))
            if not float(err[k]) <= 1.0:
                raise ValueError(
                    f"user-defined target: the hand-written gradient disagrees with central differences of the potential at "
                    f"coordinate {int(idx[k])} (gradient {float(ga[k]):.9g}, finite difference {float(fd[k]):.9g}); write the "
                    "log-density only (aehmc_logp / aehmc_glm_loglik + aehmc_glm_logprior) and let the engine differentiate it")
        return worst

    def _diag(self, C, D, nuts):
        # every array is fully written by the kernels: no memset launches
        dev = self.device
        i64 = torch.empty(2, C, dtype=torch.int64, device=dev)
        i32 = torch.empty(2, C, dtype=torch.int32, device=dev)
        out = dict(
            momentum=torch.empty(C, D, dtype=torch.float64, device=dev),
            acceptance_probability=torch.empty(C, dtype=torch.float64, device=dev),
            num_doublings=i64[0], is_turning=i32[0], is_diverging=i32[1], n_leapfrog=i64[1])
        c = _lib.CDiagnostics(**{k: v.data_ptr() for k, v in out.items()})
        out["flags"] = i32  # (is_turning, is_diverging) as one array: one conversion to bool for both
        return out, c

    def hmc_step(self, rng, eps, L, thr, q, U, g):
        C, D = q.shape
        self.ensure_workspace(C, 1)
        out, c = self._diag(C, D, False)
        self._step_call(self.lib.aehmc_hmc_step, "aehmc_hmc_step", self.ctx, C, rng.data_ptr(), float(eps),
                        int(L), float(thr), q.data_ptr(), U.data_ptr(), g.data_ptr(), ct.byref(c), self.stream)
        return out

    def hmc_sample(self, rng, eps, L, thr, n, q, U, g, keep_samples=True):
        C, D = q.shape
        self.ensure_workspace(C, 1)
        out, c = self._diag(C, D, False)
        dev = self.device
        samples = torch.empty(n, C, D, dtype=torch.float64, device=dev) if keep_samples else None
        acc = torch.empty(n, C, dtype=torch.float64, device=dev)
        div = torch.empty(n, C, dtype=torch.int32, device=dev)
        self._step_call(
            self.lib.aehmc_hmc_sample, "aehmc_hmc_sample",
            self.ctx, C, rng.data_ptr(), float(eps), int(L), float(thr), int(n), q.data_ptr(),
            U.data_ptr(), g.data_ptr(), ct.byref(c), samples.data_ptr() if keep_samples else None,
            acc.data_ptr(), div.data_ptr(), self.stream)
        out["samples"], out["acceptance_history"], out["divergence_history"] = samples, acc, div
        return out

    def nuts_step(self, rng, eps, max_exp, thr, q, U, g):
        C, D = q.shape
        self.ensure_workspace(C, max_exp)
        out, c = self._diag(C, D, True)
        self._step_call(self.lib.aehmc_nuts_step, "aehmc_nuts_step", self.ctx, C, rng.data_ptr(), float(eps),
                        int(max_exp), float(thr), q.data_ptr(), U.data_ptr(), g.data_ptr(), ct.byref(c),
                        self.stream)
        return out

    def nuts_sample(self, rng, eps, max_exp, thr, n, q, U, g, keep_samples=True):
        C, D = q.shape
        self.ensure_workspace(C, max_exp)
        out, c = self._diag(C, D, True)
        dev = self.device
        samples = torch.empty(n, C, D, dtype=torch.float64, device=dev) if keep_samples else None
        acc = torch.empty(n, C, dtype=torch.float64, device=dev)
        div = torch.empty(n, C, dtype=torch.int32, device=dev)
        total = torch.zeros(C, dtype=torch.int64, device=dev)
        self._step_call(
            self.lib.aehmc_nuts_sample, "aehmc_nuts_sample",
            self.ctx, C, rng.data_ptr(), float(eps), int(max_exp), float(thr), int(n), q.data_ptr(),
            U.data_ptr(), g.data_ptr(), ct.byref(c), samples.data_ptr() if keep_samples else None,
            acc.data_ptr(), div.data_ptr(), total.data_ptr(), self.stream)
        out["samples"], out["acceptance_history"], out["divergence_history"] = samples, acc, div
        out["n_leapfrog"] = total
        return out

    def nuts_warmup(self, rng, schedule, target_accept, max_exp, thr, q, U, g, st, cstate, imm_param):
        """The whole window-adaptation loop in one C-ABI call (no Python between warm-up steps)."""
        C, D = q.shape
        out, c = self._diag(C, D, True)
        n = len(schedule)
        stage = (ct.c_int32 * n)(*[int(s) for s, _ in schedule])
        wend = (ct.c_int32 * n)(*[int(bool(e)) for _, e in schedule])
        # bind the adaptation state's own arrays: the update kernel rewrites them in place
        self.set_metric(imm_param, D)
        self.ensure_workspace(C, max_exp)  # (after the metric: a dense one needs more work vectors)
        self._keep["eps"] = st["step_size"]
        self._check(self.lib.aehmc_set_step_sizes(self.ctx, st["step_size"].data_ptr(), C), "aehmc_set_step_sizes")
        self._step_call(self.lib.aehmc_nuts_warmup, "aehmc_nuts_warmup", self.ctx, C, rng.data_ptr(), n, stage, wend,
                        float(target_accept), int(max_exp), float(thr), q.data_ptr(), U.data_ptr(), g.data_ptr(),
                        ct.byref(c), ct.byref(cstate), self.stream)
        return out

    def hmc_warmup(self, rng, schedule, target_accept, L, thr, q, U, g, st, cstate, imm_param):
        """window_adaptation.run around an HMC kernel in one C-ABI call (no Python between warm-up steps)."""
        C, D = q.shape
        out, c = self._diag(C, D, False)
        n = len(schedule)
        stage = (ct.c_int32 * n)(*[int(s) for s, _ in schedule])
        wend = (ct.c_int32 * n)(*[int(bool(e)) for _, e in schedule])
        self.set_metric(imm_param, D)
        self.ensure_workspace(C, 1)
        self._keep["eps"] = st["step_size"]
        self._check(self.lib.aehmc_set_step_sizes(self.ctx, st["step_size"].data_ptr(), C), "aehmc_set_step_sizes")
        self._step_call(self.lib.aehmc_hmc_warmup, "aehmc_hmc_warmup", self.ctx, C, rng.data_ptr(), n, stage, wend,
                        float(target_accept), int(L), float(thr), q.data_ptr(), U.data_ptr(), g.data_ptr(),
                        ct.byref(c), ct.byref(cstate), self.stream)
        return out

    def leapfrog(self, eps, nsteps, q, p, U, g):
        C, D = q.shape
        self.ensure_workspace(C, 1)
        self._check(self.lib.aehmc_leapfrog(self.ctx, C, float(eps), int(nsteps), q.data_ptr(),
                                            p.data_ptr(), U.data_ptr(), g.data_ptr(), self.stream),
                    "aehmc_leapfrog")

    def kinetic_energy(self, p):
        C, D = p.shape
        self.ensure_workspace(C, 1)
        K = torch.empty(C, dtype=torch.float64, device=self.device)
        self._check(self.lib.aehmc_kinetic_energy(self.ctx, C, p.data_ptr(), K.data_ptr(), self.stream),
                    "aehmc_kinetic_energy")
        return K

    def is_turning(self, pl, pr, ps):
        C, D = pl.shape
        self.ensure_workspace(C, 1)
        out = torch.empty(C, dtype=torch.int32, device=self.device)
        self._check(self.lib.aehmc_is_turning(self.ctx, C, pl.data_ptr(), pr.data_ptr(), ps.data_ptr(),
                                              out.data_ptr(), self.stream), "aehmc_is_turning")
        return out.bool()

    def rng_normals(self, rng, n):
        C = rng.shape[0]
        out = torch.empty(C, n, dtype=torch.float64, device=self.device)
        self._check(self.lib.aehmc_rng_normals(self.ctx, C, rng.data_ptr(), n, out.data_ptr(), self.stream),
                    "aehmc_rng_normals")
        return out

    def rng_bernoulli(self, rng, p):
        C, n = p.shape
        out = torch.empty(C, n, dtype=torch.int32, device=self.device)
        self._check(self.lib.aehmc_rng_bernoulli(self.ctx, C, rng.data_ptr(), n, p.data_ptr(),
                                                 out.data_ptr(), self.stream), "aehmc_rng_bernoulli")
        return out

    def gemm_nt(self, A, B):
        M, K = A.shape
        N = B.shape[0]
        out = torch.empty(M, N, dtype=torch.float64, device=self.device)
        self._check(self.lib.aehmc_gemm_nt(self.ctx, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(),
                                           B.stride(0), out.data_ptr(), out.stride(0), self.stream),
                    "aehmc_gemm_nt")
        return out

    # ------------------------------------------------------------------ warm-up
    def adapt_alloc(self, C, D, full=False):
        dev, f64, i64 = self.device, torch.float64, torch.int64
        mat = (C, D, D) if full else (C, D)
        st = dict(da_step=torch.empty(C, dtype=i64, device=dev), da_x=torch.empty(C, dtype=f64, device=dev),
                  da_x_avg=torch.empty(C, dtype=f64, device=dev), da_g_avg=torch.empty(C, dtype=f64, device=dev),
                  da_mu=torch.empty(C, dtype=f64, device=dev), wc_mean=torch.empty(C, D, dtype=f64, device=dev),
                  wc_m2=torch.empty(mat, dtype=f64, device=dev), wc_n=torch.empty(C, dtype=i64, device=dev),
                  step_size=torch.empty(C, dtype=f64, device=dev), imm=torch.empty(mat, dtype=f64, device=dev),
                  sqrt_mass=torch.empty(mat, dtype=f64, device=dev))
        if full and D > 64:  # scratch of the window-end factorisation (LDS holds it up to D = 64)
            st["work"] = torch.empty(mat, dtype=f64, device=dev)
        return st, _lib.CAdaptState(full=int(bool(full)), **{k: v.data_ptr() for k, v in st.items()})

    def adapt_init(self, C, D, initial_step_size, cstate):
        self._check(self.lib.aehmc_adapt_init(self.ctx, C, D, float(initial_step_size), ct.byref(cstate),
                                              self.stream), "aehmc_adapt_init")

    def adapt_update(self, C, D, stage, window_end, last, target, p_accept, position, cstate):
        self._check(self.lib.aehmc_adapt_update(self.ctx, C, D, int(stage), int(window_end), int(last),
                                                float(target), p_accept.data_ptr(), position.data_ptr(),
                                                ct.byref(cstate), self.stream), "aehmc_adapt_update")

    def dual_averaging_update(self, target, gamma, t0, kappa, p_accept, step, x, x_avg, g_avg, mu, step_size_out):
        self._check(self.lib.aehmc_dual_averaging_update(
            self.ctx, step.numel(), float(target), float(gamma), float(t0), float(kappa), p_accept.data_ptr(),
            step.data_ptr(), x.data_ptr(), x_avg.data_ptr(), g_avg.data_ptr(), mu.data_ptr(),
            step_size_out.data_ptr() if step_size_out is not None else None, self.stream),
            "aehmc_dual_averaging_update")

    def welford_update(self, value, mean, m2, n, full):
        """algorithms.welford_covariance's update for C estimators, in place (value, mean [C,D]; m2 [C,D] | [C,D,D];
        n [C] int64)."""
        C, D = mean.shape
        self._check(self.lib.aehmc_welford_update(self.ctx, C, D, int(bool(full)), value.data_ptr(), mean.data_ptr(),
                                                  m2.data_ptr(), n.data_ptr(), self.stream), "aehmc_welford_update")

    def covariance_final(self, m2, n, D, full, shrink):
        """m2 / (n - 1) (algorithms.py:199-202), with ``shrink`` Stan's regularisation on top (mass_matrix.py:83-118)."""
        out = torch.empty_like(m2)
        self._check(self.lib.aehmc_covariance_final(self.ctx, n.numel(), int(D), int(bool(full)), int(bool(shrink)),
                                                    m2.data_ptr(), n.data_ptr(), out.data_ptr(), self.stream),
                    "aehmc_covariance_final")
        return out

    def profile_enable(self, on=True):
        self._check(self.lib.aehmc_profile_enable(self.ctx, int(on)), "aehmc_profile_enable")

    def profile_read(self):
        ms, n, fl = ct.c_double(), ct.c_int64(), ct.c_double()
        self._check(self.lib.aehmc_profile_read(self.ctx, ct.byref(ms), ct.byref(n), ct.byref(fl)),
                    "aehmc_profile_read")
        return ms.value, n.value, fl.value


_engines = {}


def get_engine(device=None) -> Engine:
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}") \
        if torch.cuda.is_available() else None
    if dev is None:
        return Engine(device)  # raises the loud error
    if dev not in _engines:
        _engines[dev] = Engine(dev)
    return _engines[dev]


def rng_to_device(sites: np.ndarray, device) -> torch.Tensor:
    """uint64 [C,n,4] -> int64 device tensor with the same bits."""
    return torch.from_numpy(np.ascontiguousarray(sites).view(np.int64)).to(device)
