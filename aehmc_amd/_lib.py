"""ctypes binding of libaehmc_hip.so (include/aehmc_hip.h).  Fails loudly when the HIP
library is missing: there is no CPU path in this package."""
from __future__ import annotations

import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AEHMC_AMD_LIB: developer override (e.g. the instrumented `make timing` build); never a CPU path
LIB_PATH = os.environ.get("AEHMC_AMD_LIB") or os.path.join(_HERE, "libaehmc_hip.so")
_lib = None


class CTarget(ct.Structure):
    _fields_ = [("kind", ct.c_int32), ("reserved", ct.c_int32), ("D", ct.c_int64),
                ("mu", ct.c_void_p), ("sigma", ct.c_void_p), ("prec", ct.c_void_p),
                ("X", ct.c_void_p), ("y", ct.c_void_p), ("N", ct.c_int64)]


class CMetric(ct.Structure):
    _fields_ = [("ndim", ct.c_int32), ("per_chain", ct.c_int32), ("D", ct.c_int64),
                ("imm", ct.c_void_p), ("sqrt_mass", ct.c_void_p), ("n_chains", ct.c_int64)]


class CAdaptState(ct.Structure):
    _fields_ = [(n, ct.c_void_p) for n in ("da_step", "da_x", "da_x_avg", "da_g_avg", "da_mu",
                                           "wc_mean", "wc_m2", "wc_n", "step_size", "imm",
                                           "sqrt_mass")] + [("full", ct.c_int32), ("reserved", ct.c_int32),
                                                            ("work", ct.c_void_p)]


class CDiagnostics(ct.Structure):
    _fields_ = [("momentum", ct.c_void_p), ("acceptance_probability", ct.c_void_p),
                ("num_doublings", ct.c_void_p), ("is_turning", ct.c_void_p),
                ("is_diverging", ct.c_void_p), ("n_leapfrog", ct.c_void_p)]


# every symbol include/aehmc_hip.h declares: name -> (restype, argtypes)
_P, _I64, _D, _I = ct.c_void_p, ct.c_int64, ct.c_double, ct.c_int
SYMBOLS = {
    "aehmc_create": (_I, [ct.POINTER(_P), _I]),
    "aehmc_destroy": (_I, [_P]),
    "aehmc_last_error": (ct.c_char_p, [_P]),
    "aehmc_set_target": (_I, [_P, ct.POINTER(CTarget)]),
    "aehmc_set_custom_target": (_I, [_P, ct.c_char_p, _I64, ct.POINTER(_P), ct.c_int32, ct.c_char_p]),
    "aehmc_set_custom_joint_target": (_I, [_P, ct.c_char_p, _I64, ct.POINTER(_P), ct.c_int32, ct.c_char_p]),
    "aehmc_set_rtc_cache": (_I, [_P, ct.c_char_p]),
    "aehmc_rtc_stats": (_I, [_P, ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int64)]),
    "aehmc_set_custom_glm_target": (_I, [_P, ct.c_char_p, _I64, _I64, _P, _P, ct.POINTER(_P), ct.c_int32, ct.c_char_p]),
    "aehmc_set_metric": (_I, [_P, ct.POINTER(CMetric)]),
    "aehmc_set_step_sizes": (_I, [_P, _P, _I64]),
    "aehmc_metric_sqrt": (_I, [_P, ct.c_int32, _I64, _P, _P, _P]),
    "aehmc_metric_sqrt_per_chain": (_I, [_P, _I64, _I64, _P, _P, _P]),
    "aehmc_adapt_init": (_I, [_P, _I64, _I64, _D, ct.POINTER(CAdaptState), _P]),
    "aehmc_adapt_update": (_I, [_P, _I64, _I64, ct.c_int32, ct.c_int32, ct.c_int32, _D, _P, _P,
                                ct.POINTER(CAdaptState), _P]),
    "aehmc_dual_averaging_update": (_I, [_P, _I64, _D, _D, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aehmc_welford_update": (_I, [_P, _I64, _I64, ct.c_int32, _P, _P, _P, _P, _P]),
    "aehmc_covariance_final": (_I, [_P, _I64, _I64, ct.c_int32, ct.c_int32, _P, _P, _P, _P]),
    "aehmc_set_option": (_I, [_P, ct.c_char_p, _I64]),
    "aehmc_workspace_bytes": (_I64, [_P, _I64, _I64]),
    "aehmc_set_workspace": (_I, [_P, _P, _I64]),
    "aehmc_new_state": (_I, [_P, _I64, _P, _P, _P, _P]),
    "aehmc_hmc_step": (_I, [_P, _I64, _P, _D, _I64, _D, _P, _P, _P, ct.POINTER(CDiagnostics), _P]),
    "aehmc_hmc_sample": (_I, [_P, _I64, _P, _D, _I64, _D, _I64, _P, _P, _P, ct.POINTER(CDiagnostics),
                              _P, _P, _P, _P]),
    "aehmc_nuts_step": (_I, [_P, _I64, _P, _D, _I64, _D, _P, _P, _P, ct.POINTER(CDiagnostics), _P]),
    "aehmc_nuts_sample": (_I, [_P, _I64, _P, _D, _I64, _D, _I64, _P, _P, _P, ct.POINTER(CDiagnostics),
                               _P, _P, _P, _P, _P]),
    "aehmc_nuts_warmup": (_I, [_P, _I64, _P, _I64, _P, _P, _D, _I64, _D, _P, _P, _P, ct.POINTER(CDiagnostics),
                               ct.POINTER(CAdaptState), _P]),
    "aehmc_hmc_warmup": (_I, [_P, _I64, _P, _I64, _P, _P, _D, _I64, _D, _P, _P, _P, ct.POINTER(CDiagnostics),
                              ct.POINTER(CAdaptState), _P]),
    "aehmc_leapfrog": (_I, [_P, _I64, _D, _I64, _P, _P, _P, _P, _P]),
    "aehmc_kinetic_energy": (_I, [_P, _I64, _P, _P, _P]),
    "aehmc_is_turning": (_I, [_P, _I64, _P, _P, _P, _P, _P]),
    "aehmc_rng_normals": (_I, [_P, _I64, _P, _I64, _P, _P]),
    "aehmc_rng_bernoulli": (_I, [_P, _I64, _P, _I64, _P, _P, _P]),
    "aehmc_gemm_nt": (_I, [_P, _I64, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _P]),
    "aehmc_profile_enable": (_I, [_P, _I]),
    "aehmc_profile_read": (_I, [_P, ct.POINTER(_D), ct.POINTER(_I64), ct.POINTER(_D)]),
    "aehmc_synchronize": (_I, [_P, _P]),
}


def load():
    """Load the HIP library; raise (never fall back) if it is absent or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C aehmc_amd/csrc`); aehmc_amd has no CPU fallback")
        if not os.environ.get("AEHMC_AMD_LIB") and not os.environ.get("AEHMC_AMD_ALLOW_STALE"):
            from . import _build
            if not _build.is_current():  # a binary of OTHER sources must not pass for this tree's
                raise RuntimeError(
                    f"{LIB_PATH} was not built from the sources in this tree (source hash "
                    f"{_build.source_hash()[:16]}, stamped {str(_build.stamped_hash())[:16]}): rebuild it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ct.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
