"""ctypes front-end of oracle/libaehmc_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Mirrors the call shapes of the product engine so that parity tests read the same on both
sides: arrays are ``[C, D]`` float64, per-chain RNG state is ``[C, n_sites, 4]`` uint64
(PCG64 ``state_hi, state_lo, inc_hi, inc_lo`` per call site, scheme A).
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

T_STD_NORMAL, T_ISO_GAUSSIAN, T_DIAG_GAUSSIAN, T_DENSE_MVN, T_LINREG = range(5)


class _Target(ct.Structure):
    _fields_ = [("kind", ct.c_int32), ("pad", ct.c_int32), ("D", ct.c_int64),
                ("mu", ct.c_void_p), ("sigma", ct.c_void_p), ("prec", ct.c_void_p),
                ("X", ct.c_void_p), ("y", ct.c_void_p), ("N", ct.c_int64)]


class _Metric(ct.Structure):
    _fields_ = [("ndim", ct.c_int32), ("pad", ct.c_int32), ("D", ct.c_int64),
                ("imm", ct.c_void_p), ("sqrt_mass", ct.c_void_p)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libaehmc_oracle.so")
    src = os.path.join(_HERE, "c", "aehmc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libaehmc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # AEHMC_ORACLE_LIB: another build of the same source (the ASan / UBSan one of `make asan`)
        _LIB = ct.CDLL(os.environ.get("AEHMC_ORACLE_LIB") or build())
        _LIB.ao_kinetic_energy.restype = ct.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(ct.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------- RNG seeds
def site_states(seeds, n_sites: int, first_site: int = 0) -> np.ndarray:
    """Scheme A: chain c, site k -> PCG64(SeedSequence(seed_c).spawn(...)[first_site+k]).
    Returns uint64 [C, n_sites, 4] = (state_hi, state_lo, inc_hi, inc_lo)."""
    seeds = np.atleast_1d(np.asarray(seeds))
    out = np.empty((len(seeds), n_sites, 4), dtype=np.uint64)
    m64 = (1 << 64) - 1
    for c, s in enumerate(seeds):
        children = np.random.SeedSequence(int(s)).spawn(first_site + n_sites)
        for k in range(n_sites):
            st = np.random.PCG64(children[first_site + k]).state["state"]
            out[c, k] = (st["state"] >> 64, st["state"] & m64, st["inc"] >> 64, st["inc"] & m64)
    return out


# ---------------------------------------------------------------------------- holders
class Target:
    def __init__(self, kind, D, mu=None, sigma=None, prec=None, X=None, y=None):
        self.kind, self.D = kind, int(D)
        self.mu = _f64(mu) if mu is not None else None
        self.sigma = _f64(sigma) if sigma is not None else None
        self.prec = _f64(prec) if prec is not None else None
        self.X = _f64(X) if X is not None else None
        self.y = _f64(y) if y is not None else None
        self.c = _Target(kind, 0, self.D, _p(self.mu), _p(self.sigma), _p(self.prec),
                         _p(self.X), _p(self.y), 0 if self.X is None else len(self.X))


class Metric:
    """gaussian_metric (metrics.py:44-63): ndim 0/1/2 by the shape of imm."""

    def __init__(self, inverse_mass_matrix, D):
        imm = np.asarray(inverse_mass_matrix, dtype=np.float64)
        if imm.ndim > 2:
            raise ValueError(
                f"Expected a mass matrix of dimension 1 (diagonal) or 2, got {imm.ndim}")
        self.ndim, self.D = imm.ndim, int(D)
        if imm.ndim < 2:
            self.imm = _f64(np.atleast_1d(imm))
            self.sqrt_mass = _f64(np.sqrt(np.reciprocal(self.imm)))
        else:
            import scipy.linalg
            self.imm = _f64(imm)
            L = np.linalg.cholesky(imm)
            self.sqrt_mass = _f64(scipy.linalg.solve_triangular(
                L, np.eye(imm.shape[0]), lower=True, trans=1))
        self.c = _Metric(self.ndim, 0, self.D, _p(self.imm), _p(self.sqrt_mass))


# ---------------------------------------------------------------------------- calls
def new_state(target: Target, q):
    q = _f64(np.atleast_2d(q)).reshape(-1, target.D)
    C = q.shape[0]
    U = np.empty(C)
    g = np.empty_like(q)
    lib().ao_new_state(ct.byref(target.c), ct.c_int64(C), _p(q), _p(U), _p(g))
    return q, U, g


def hmc_step(target, metric, rng, eps, L, q, U, g, thr=1000.0, nthreads=1):
    """One HMC transition for C chains; state arrays are updated in place; rng [C,2,4]."""
    C = q.shape[0]
    p = np.empty_like(q)
    acc = np.empty(C)
    div = np.empty(C, dtype=np.int32)
    accepted = np.empty(C, dtype=np.int32)
    lib().ao_hmc_step(ct.byref(target.c), ct.byref(metric.c), ct.c_int64(C), _p(rng),
                      ct.c_double(eps), ct.c_int64(L), ct.c_double(thr), _p(q), _p(U), _p(g),
                      _p(p), _p(acc), _p(div), _p(accepted), ct.c_int32(nthreads))
    return dict(momentum=p, acceptance_probability=acc, is_diverging=div.astype(bool),
                accepted=accepted.astype(bool), n_leapfrog=np.full(C, L, dtype=np.int64))


def nuts_step(target, metric, rng, eps, q, U, g, max_exp=10, thr=1000.0, nthreads=1):
    """One NUTS transition for C chains; state arrays updated in place; rng [C,4,4]."""
    C = q.shape[0]
    p = np.empty_like(q)
    acc = np.empty(C)
    nd = np.empty(C, dtype=np.int64)
    turn = np.empty(C, dtype=np.int32)
    div = np.empty(C, dtype=np.int32)
    nl = np.empty(C, dtype=np.int64)
    lib().ao_nuts_step(ct.byref(target.c), ct.byref(metric.c), ct.c_int64(C), _p(rng),
                       ct.c_double(eps), ct.c_int64(max_exp), ct.c_double(thr), _p(q), _p(U),
                       _p(g), _p(p), _p(acc), _p(nd), _p(turn), _p(div), _p(nl),
                       ct.c_int32(nthreads))
    return dict(momentum=p, acceptance_probability=acc, num_doublings=nd,
                is_turning=turn.astype(bool), is_diverging=div.astype(bool), n_leapfrog=nl)


def set_experiment(alt_quirk1=False, alt_quirk2=False):
    """Experiment switches of the C restatement (see aehmc_oracle.c); (False, False) is the
    reference's semantics.  Used only by tests/test_nuts_quirks.py."""
    lib().ao_set_experiment(ct.c_int32(int(alt_quirk1)), ct.c_int32(int(alt_quirk2)))


def leapfrog(target, metric, eps, nsteps, q, p, U, g):
    lib().ao_leapfrog(ct.byref(target.c), ct.byref(metric.c), ct.c_int64(q.shape[0]),
                      ct.c_double(eps), ct.c_int64(nsteps), _p(q), _p(p), _p(U), _p(g))


def rng_normals(state, n):
    out = np.empty(n)
    lib().ao_rng_normals(_p(state), ct.c_int64(n), _p(out))
    return out


def rng_doubles(state, n):
    out = np.empty(n)
    lib().ao_rng_doubles(_p(state), ct.c_int64(n), _p(out))
    return out


def rng_bernoulli(state, p):
    p = _f64(p)
    out = np.empty(len(p), dtype=np.int32)
    lib().ao_rng_bernoulli(_p(state), ct.c_int64(len(p)), _p(p), _p(out))
    return out


def find_storage_indices(step):
    mn, mx = ct.c_int64(), ct.c_int64()
    lib().ao_find_storage_indices(ct.c_int64(step), ct.byref(mn), ct.byref(mx))
    return mn.value, mx.value


def is_iterative_turning(metric, ckp, cks, mn, mx, psum, p):
    ckp, cks, psum, p = _f64(ckp), _f64(cks), _f64(np.atleast_1d(psum)), _f64(np.atleast_1d(p))
    return bool(lib().ao_is_iterative_turning(ct.byref(metric.c), ct.c_int64(len(ckp)), _p(ckp),
                                              _p(cks), ct.c_int64(mn), ct.c_int64(mx),
                                              _p(psum), _p(p)))


def is_turning(metric, pl, pr, ps):
    pl, pr, ps = (_f64(np.atleast_1d(x)) for x in (pl, pr, ps))
    return bool(lib().ao_is_turning(ct.byref(metric.c), _p(pl), _p(pr), _p(ps)))


def kinetic_energy(metric, p):
    p = _f64(np.atleast_1d(p))
    return float(lib().ao_kinetic_energy(ct.byref(metric.c), _p(p)))
