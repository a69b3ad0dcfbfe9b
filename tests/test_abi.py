"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/aehmc_hip.h declares, and the product package has no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from aehmc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "aehmc_hip.h")).read()
    declared = set(re.findall(r"\b(aehmc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_enums_match_header():
    from aehmc_amd import targets
    hdr = open(os.path.join(ROOT, "include", "aehmc_hip.h")).read()
    for name, val in [("STD_NORMAL", targets.T_STD_NORMAL), ("ISO_GAUSSIAN", targets.T_ISO_GAUSSIAN),
                      ("DIAG_GAUSSIAN", targets.T_DIAG_GAUSSIAN), ("DENSE_MVN", targets.T_DENSE_MVN),
                      ("LINREG", targets.T_LINREG)]:
        assert re.search(rf"AEHMC_T_{name}\s*=\s*{val}\b", hdr)


def test_product_never_imports_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "aehmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libaehmc_oracle" not in src and "np_oracle" not in src, f


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from aehmc_amd import hmc, targets
    from aehmc_amd.engine import EngineError
    with pytest.raises(EngineError):
        hmc.new_state(0.0, targets.StdNormal())


def test_random_stream_scheme_a():
    import numpy as np
    from aehmc_amd import RandomStream
    s = RandomStream(seed=0)
    st = s.sites(4)
    g = np.random.default_rng(np.random.SeedSequence(0).spawn(4)[2]).bit_generator.state["state"]
    assert int(st[0, 2, 0]) == g["state"] >> 64 and int(st[0, 2, 2]) == g["inc"] >> 64
    # later kernels continue the spawn sequence (hmc after nuts on one srng)
    st2 = s.sites(2)
    g5 = np.random.default_rng(np.random.SeedSequence(0).spawn(6)[5]).bit_generator.state["state"]
    assert int(st2[0, 1, 1]) == g5["state"] & (2**64 - 1)
