"""bench.py prints ONE JSON line with the fields the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _run(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_c2_line():
    d = _run("--config", "c2", "--steps", "3", "--warmup", "1")
    assert all(k in d for k in REQUIRED)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and d["value"] > 1e8
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_c3_small_line():
    # the headline workload at a reduced size (same code path: dense MFMA GEMMs, stream-K, compaction)
    d = _run("--steps", "2", "--warmup", "1", "--chains", "256", "--dim", "1024", "--no-cpu-baseline")
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["launches"] > 0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
    assert d["config"]["leapfrogs_per_step"] > 0 and d["value"] > 0


def test_bench_gpus2_starts_two_ranks():
    """`python bench.py --gpus 2` (no launcher) must start two ranks itself and report n_gpus == 2.
    On a 1-GPU box both ranks share cuda:0 (AEHMC_BENCH_ONE_DEVICE=1) and the gather runs over gloo
    (RCCL refuses two ranks on one device); on the driver's node it is one rank per GPU over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AEHMC_BENCH_ONE_DEVICE="1", AEHMC_DIST_BACKEND="gloo")
    d = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "64", "--dim", "256",
             "--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["chains_total"] == 128
    assert d["config"]["gather"]["bytes"] == 64 * 256 * 8 and d["value"] > 0
