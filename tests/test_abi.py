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


def test_random_stream_from_state_round_trip():
    """RandomStream.from_state hands the saved generator states to the NEXT kernel and keeps the spawn sequence
    of the seeds where an unbroken session would be (README.md:49-51: the reference threads RNG state through
    `updates`); without the seeds it can resume one kernel only."""
    import numpy as np
    from aehmc_amd import RandomStream
    s = RandomStream(seed=0)
    nuts_sites, hmc_sites = s.sites(4), s.sites(2)
    r = RandomStream.from_state(nuts_sites.view(np.int64), seed=0)  # int64 bit patterns, as a device tensor holds them
    assert not r.batched and r.num_chains == 1
    assert np.array_equal(r.sites(4), nuts_sites) and np.array_equal(r.sites(2), hmc_sites)
    many = RandomStream(seeds=[3, 4, 5])
    st = many.sites(2)
    lone = RandomStream.from_state(st)
    assert lone.batched and lone.num_chains == 3 and np.array_equal(lone.sites(2), st)
    with pytest.raises(ValueError, match="without its seeds"):
        lone.sites(2)
    with pytest.raises(ValueError, match="call sites"):
        RandomStream.from_state(st, seeds=[3, 4, 5]).sites(4)
    with pytest.raises(ValueError, match="saved chains"):
        RandomStream.from_state(st, seeds=[3, 4])
    with pytest.raises(ValueError, match=r"\[C, n_sites, 4\]"):
        RandomStream.from_state(st[0])


def test_window_adaptation_schedule_and_hmc_argument():
    """build_schedule is the reference's (tests/test_adaptation.py:9-22 pins it in test_adaptation_oracle.py); an
    HMC kernel without its trajectory length is refused before anything touches the GPU."""
    from aehmc_amd import window_adaptation
    sched = window_adaptation.build_schedule(1000)
    assert len(sched) == 1000 and sched[0] == (0, False) and sched[75] == (1, False)
    assert sum(1 for _, e in sched if e) == 5 and sched[-1] == (0, False)

    def fake_hmc(state, eps, imm, L):
        raise AssertionError("not reached")
    fake_hmc._hmc = {}
    with pytest.raises(ValueError, match="num_integration_steps"):
        window_adaptation.run(fake_hmc, None, 10)


def test_every_engine_option_is_documented_in_the_header():
    """aehmc_set_option accepts the names the header's option table lists, and no others."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "aehmc_amd", "csrc", "engine.hip")).read()
    hdr = open(os.path.join(root, "include", "aehmc_hip.h")).read()
    accepted = set(re.findall(r'strcmp\(name, "([a-z_]+)"\)', src))
    table = hdr[hdr.index("/* engine options (name, default):"):hdr.index("int aehmc_set_option")]
    documented = set(re.findall(r'^ \*  "([a-z_]+)"', table, flags=re.M))
    assert accepted and accepted == documented, accepted ^ documented


def test_build_is_keyed_by_source_content_not_mtime(monkeypatch):
    """__graft_entry__.build() recompiles when the hash of csrc/* + include/* differs from the one stamped beside the
    .so -- a comment edited in a header is enough, file times are not consulted -- and the loader refuses a binary
    that was built from other sources."""
    import subprocess

    from aehmc_amd import _build, _lib
    assert _build.is_current(), "run build() first"
    calls = []
    monkeypatch.setattr(subprocess, "check_call", lambda cmd, *a, **k: calls.append(cmd))
    assert _build.build() is False and calls == []  # up to date: nothing compiled
    hdr = os.path.join(ROOT, "include", "aehmc_hip.h")
    saved, stamp = open(hdr).read(), open(_build.STAMP).read()
    st = os.stat(hdr)
    try:
        with open(hdr, "a") as f:
            f.write("/* edited */\n")
        os.utime(hdr, (st.st_atime, st.st_mtime))  # same mtime as before: make alone would not rebuild
        assert not _build.is_current()
        monkeypatch.delenv("AEHMC_AMD_LIB", raising=False)
        monkeypatch.setattr(_lib, "_lib", None)
        with pytest.raises(RuntimeError, match="was not built from the sources"):
            _lib.load()
        assert _build.build() is True
        assert calls and calls[0][:3] == ["make", "-C", _build.CSRC] and "-B" in calls[0]
    finally:
        open(hdr, "w").write(saved)
        os.utime(hdr, (st.st_atime, st.st_mtime))
        open(_build.STAMP, "w").write(stamp)
    assert _build.is_current()


def test_large_host_arrays_are_keyed_by_full_content():
    """An in-place edit of ONE element of a > 64 MB numpy parameter changes the cache key (round 3 hashed a 1/256
    sample there and missed it); the sampled key exists only as an explicit opt-in."""
    import warnings

    import numpy as np
    from aehmc_amd import engine
    a = np.zeros((3000, 3000))  # 72 MB
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        k0 = engine._param_key(a)
        a[1234, 1235] = 1e-300  # flat index 3703235: not a multiple of 256, not in the first or last MB
        k1 = engine._param_key(a)
        assert k0 != k1
        # lists / float32 / non-contiguous inputs are keyed by content too: equal content, equal key
        assert engine._param_key(a[::2, ::2]) == engine._param_key(np.ascontiguousarray(a[::2, ::2]))
        try:
            engine.SAMPLED_HASH_ABOVE = 64 << 20
            s0 = engine._param_key(a)
            a[1234, 1237] = 1e-300
            assert engine._param_key(a) == s0  # the opt-in's documented blind spot
        finally:
            engine.SAMPLED_HASH_ABOVE = None


def test_cpu_tensors_are_keyed_by_content():
    """A CPU torch tensor edited through a numpy view of its storage keeps its version counter:
    identity + version would serve the stale upload.  CPU tensors are keyed by content like numpy arrays."""
    import torch
    from aehmc_amd import engine
    t = torch.zeros(64, dtype=torch.float64)
    k0, v0 = engine._param_key(t), t._version
    t.numpy()[3] = 2.5
    assert t._version == v0            # the edit is invisible to the version counter ...
    assert engine._param_key(t) != k0  # ... and visible to the key
    assert engine._param_key(t) == engine._param_key(t.clone())  # equal content, equal key
    assert engine._param_key(torch.zeros(4, dtype=torch.float32))[0] == "c"


def test_random_stream_from_state_for_a_later_kernel_and_dtype_checks():
    """Resuming a kernel that was NOT the first one built from its stream: `n_spawned` puts the spawn sequence where the
    unbroken session had it, so kernels built after the resumed one get the call sites they would have got; states of
    another dtype (floats, 32-bit ints, out-of-range Python ints) are refused instead of truncated."""
    import numpy as np
    from aehmc_amd import RandomStream
    s = RandomStream(seed=3)
    first, second, third = s.sites(4), s.sites(2), s.sites(4)
    assert s.n_spawned == 10
    r = RandomStream.from_state(second, seed=3, n_spawned=4)  # resume the SECOND kernel
    assert np.array_equal(r.sites(2), second) and np.array_equal(r.sites(4), third) and r.n_spawned == 10
    wrong = RandomStream.from_state(second, seed=3)  # without n_spawned the next kernel would get the first one's sites
    wrong.sites(2)
    assert not np.array_equal(wrong.sites(4), third)
    mixed = [[[-1, 2 ** 63 + 5, 7, 9], [1, 2, 3, 4]]]  # negative (int64 view) and >= 2**63 (uint64) in one nested list
    got = RandomStream.from_state(mixed).sites(2)
    assert got.dtype == np.uint64 and int(got[0, 0, 0]) == 2 ** 64 - 1 and int(got[0, 0, 1]) == 2 ** 63 + 5
    for bad in (np.zeros((1, 2, 4)), np.zeros((1, 2, 4), dtype=np.int32), [[[2 ** 64, 0, 0, 0], [0, 0, 0, 0]]]):
        with pytest.raises(ValueError):
            RandomStream.from_state(bad)


def test_bench_line_fits_the_drivers_stdout_tail(capsys):
    """The driver keeps ~8 kB of bench.py's stdout: the whole JSON line -- all secondary entries with their values and
    roofline fractions -- has to fit.  Round 4's line (13 kB with its `note` strings: profiles/r4/bench_default.json)
    goes through bench.emit() and must come out below the limit with every number still in it."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r4", "bench_default.json")))
    assert len(json.dumps(full)) > bench.LINE_LIMIT  # (the fixture is a line that did NOT fit)
    line = bench.emit(full)
    assert len(line) < bench.LINE_LIMIT, len(line)
    got = json.loads(line)
    assert capsys.readouterr().out.strip() == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k  # the contract's keys survive (vs_baseline: null stays)
    assert got["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert [e["config"] for e in got["secondary"]] == [e["config"] for e in full["secondary"]]
    for e, f in zip(got["secondary"], full["secondary"]):
        assert e["value"] == pytest.approx(f["value"], rel=1e-5) and "frac" in e["roofline"]
    assert "note" not in line
    assert len(bench.emit(full, verbose=True)) > bench.LINE_LIMIT  # --verbose keeps the prose
