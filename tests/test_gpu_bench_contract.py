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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _run.last_line = lines[0]
    return json.loads(lines[0])


def test_bench_c2_line():
    d = _run("--config", "c2", "--steps", "3", "--warmup", "1")
    assert all(k in d for k in REQUIRED)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and d["value"] > 0
    r, c = d["roofline"], d["cpu_baseline"]
    # the kernel is bound by fp64 VALU issue.  `frac` is against SURVEY.md 8d's fixed ceiling (9 D flop per leapfrog over
    # the fp64 vector peak: efficiency); the ceiling at the kernel's own counted instructions per transition (offline
    # rocprofv3 counters of THIS build of the library) is reported beside it as valu.issue_frac, with HBM; counters of
    # another build are dropped and the line says why
    assert r["bound"] == "valu" and len(d["config"]["lib_sha256"]) == 64
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and 0 < r["frac"] < 1 and r["peak"] == 78.6
    assert r["achieved"] == pytest.approx(d["value"] * 9 * 100 / 1e12, rel=1e-4)
    if r.get("counters_dropped"):
        assert r.get("traffic") is None and "valu" not in r and r.get("traffic_source") is None
    else:
        assert 0 < r["valu"]["issue_frac"] < 1 and r["valu"]["issue_frac"] > r["frac"]
        assert r["traffic_source"].startswith("profiles/")
        assert r["hbm"]["frac"] < 0.05
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_counters_of_another_build_are_dropped(tmp_path):
    """Every counter summary under profiles/ carries the sha256 of the libaehmc_hip.so it was measured on; bench.py uses
    it only when that is the library it has loaded.  A summary with another hash: peak / frac / traffic are blanked and
    `counters_dropped` says why; the same summary with the right hash: the fraction is back."""
    from aehmc_amd import _build
    derived = {"valu_issue_ceiling_leapfrogs_per_s_at_this_instruction_count": 2.2e10, "hbm_bytes_per_launch": 2.4e7,
               "valu_instructions_per_wave_per_transition": 870.0, "leapfrog_fp64_instructions_per_transition": 384,
               "valu_busy_fraction_of_kernel_time_at_2.4GHz": 0.7}
    (tmp_path / "r4").mkdir()
    env = dict(os.environ, AEHMC_PROFILES_DIR=str(tmp_path))
    (tmp_path / "r4" / "c2_pmc_summary.json").write_text(json.dumps({"lib_sha256": "0" * 64, "derived": derived}))
    r = _run("--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", env=env)["roofline"]
    assert r.get("traffic") is None and "valu" not in r and 0 < r["frac"] < 1  # (the fixed-ceiling fraction needs no counters)
    assert "another build" in r["counters_dropped"] and "0000000000000000" in r["counters_dropped"]
    (tmp_path / "r4" / "c2_pmc_summary.json").write_text(json.dumps({"lib_sha256": _build.library_hash(), "derived": derived}))
    r = _run("--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", env=env)["roofline"]
    assert r.get("counters_dropped") is None and r["valu"]["issue_ceiling"] == 2.2e10
    assert r["valu"]["issue_frac"] == pytest.approx(r["achieved"] * 1e12 / 900 / 2.2e10, rel=1e-4)
    assert r["traffic"] == 2.4e7 and r["traffic_source"].startswith("profiles/")


def test_bench_c3_small_line():
    # the headline workload at a reduced size (same code path: dense MFMA GEMMs, stream-K, compaction)
    d = _run("--steps", "2", "--warmup", "1", "--chains", "256", "--dim", "1024", "--no-cpu-baseline")
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["launches"] > 0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and 0 < r["frac"] < 1
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
    assert d["config"]["gather"]["rows_match_ranks_bitwise"] is True


def test_bench_two_gpus_over_rccl():
    """Self-skipping: on a box with >= 2 GPUs `bench.py --gpus 2 --config c4` runs one rank per GPU over the REAL
    backend (RCCL over xGMI): both ranks seen, the gather moves the second rank's rows to rank 0 bit for bit, its time
    is reported.  (The pool's 1-GPU boxes skip it; the driver's multi-GPU node exercises the N > 1 path.)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AEHMC_BENCH_ONE_DEVICE",
                                                            "AEHMC_DIST_BACKEND")}
    d = _run("--gpus", "2", "--config", "c4", "--dim", "2048", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["ranks_seen"] == 2
    g = d["config"]["gather"]
    assert g["backend"] == "nccl" and g["rows_match_ranks_bitwise"] is True and g["ms"] > 0
    assert g["bytes"] == 16384 * 2048 * 8 and d["value"] > 0


def test_bench_c4_and_c5_two_rank_dry_runs():
    """The sharded configs as the driver's multi-GPU run starts them, two ranks on this box's one GPU:
    c4 = 32768 chains split over the ranks (strong scaling), c5 = the regression with warm-up."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AEHMC_BENCH_ONE_DEVICE="1", AEHMC_DIST_BACKEND="gloo")
    d = _run("--gpus", "2", "--config", "c4", "--steps", "1", "--warmup", "1", "--dim", "128", "--no-cpu-baseline",
             env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["chains_total"] == 32768
    assert d["config"]["ranks_seen"] == 2 and d["value"] > 0 and d["roofline"]["bound"] == "mfma"
    d = _run("--gpus", "2", "--config", "c5", "--steps", "5", "--warmup", "40", "--chains", "64", "--no-cpu-baseline",
             env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["chains_total"] == 128 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "valu" and r["kernel"] == "k_nuts_linreg" and r["launches"] == 1
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and 0 < r["frac"] < 1


def test_bench_default_line_has_its_good_secondary_entries():
    """The default line (c3 at full size) carries the diagonal-mass NUTS / HMC numbers, the mid-size dense problems and
    c2, c5, c1 as `secondary`: none of them may have degraded into an {"error": ...} entry."""
    d = _run("--steps", "2", "--no-cpu-baseline")
    assert len(_run.last_line) < 8000  # the whole line fits the ~8 kB of stdout the driver keeps
    assert all(k in d for k in REQUIRED) and d["roofline"]["bound"] == "mfma"
    assert d["config"]["dim"] == 10_000 and d["config"]["chains_total"] == 4096
    sec = d["secondary"]
    assert [e["config"] for e in sec] == ["diag-nuts", "diag-hmc", "diag-hmc-fp_contract", "dense-nuts-d100", "dense-nuts-d200",
                                          "pc-dense-nuts-d200", "custom-student-t-nuts-d5000", "python-funnel-nuts-d1000", "custom-logistic-nuts-n100000", "c2", "c2-fp_contract", "c5", "c1"]
    for e in sec:
        assert "error" not in e and e["value"] > 0, e
    by = {e["config"]: e for e in sec}
    assert by["c5"]["roofline"]["bound"] == "valu" and by["c2"]["roofline"]["bound"] == "valu"
    assert by["c1"]["roofline"]["bound"] == "latency"
    for k in ("dense-nuts-d100", "dense-nuts-d200"):  # block-resident kernels: one launch per sample() call
        r = by[k]["roofline"]
        assert r["bound"] == "mfma" and r["launches"] == 1 and 0 < r["frac"] < 1
    r = by["pc-dense-nuts-d200"]["roofline"]  # one dense metric per chain: bound by streaming the matrices
    assert r["bound"] == "hbm" and r["launches"] == 1 and 0 < r["frac"] < 1
    # the fast-arithmetic mode says that it is not the bit-exact mode (that it is faster is a perf check: test_gpu_perf.py)
    assert "fp_contract=1" in by["c2-fp_contract"].get("workload", "fp_contract=1") and "fp_contract" not in by["c2"].get("workload", "")
    assert "workload" in by["c2"]  # (the line sheds `traffic_source` and long counter lists before it sheds the workload texts)


def test_parallel_collectives_on_rccl_one_rank():
    """The collectives of aehmc_amd/parallel.py on the REAL backend (nccl == RCCL) with a one-rank
    group on this box's GPU: process-group creation with device_id, barrier(device_ids), the
    all-reduces of the timing / leapfrog totals and the rank-0 gather (ragged and even) run through
    RCCL's code paths rather than gloo's.  (World sizes > 1 need more GPUs than this box has: the
    driver's 8-GPU run covers them; the sharding logic itself is tested with gloo on CPU.)"""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from aehmc_amd import parallel
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl"
parallel.barrier(dev)
assert parallel.max_over_ranks(1.25, dev) == 1.25 and parallel.sum_over_ranks(7, dev) == 7
x = torch.arange(12, dtype=torch.float64, device=dev).reshape(6, 2)
g = parallel.gather_samples(x)
assert torch.equal(g, x) and g.device == x.device
g = parallel.gather_samples(x, dst=None)
assert torch.equal(g, x)
assert parallel.shard_chains(10) == (0, 10)
dist.destroy_process_group()
print("RCCL-OK")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code, ROOT, str(_free_port())], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stderr[-3000:]


def test_bench_one_rank_under_the_launcher_uses_rccl():
    """bench.py as the driver launches it (torch.distributed.run, here with one process): the rank
    initialises RCCL, and the line reports the backend it gathered over."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--chains", "64", "--dim", "256",
                          "--no-cpu-baseline", "--no-secondary"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1 and d["value"] > 0
    assert d["config"]["gather"]["backend"] == "nccl"


def test_integration_md_binding_snippet_runs():
    """The ctypes stub INTEGRATION.md shows a reference maintainer is executed verbatim: it must load the
    library, run one NUTS transition through the C-ABI and leave finite results (a doc that drifts from
    include/aehmc_hip.h -- a struct field added, a signature changed -- fails here)."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('"libaehmc_hip.so"', repr(os.path.join(ROOT, "aehmc_amd", "libaehmc_hip.so")))  # (not on the loader path here)
    prog = code + ("\ntorch.cuda.synchronize()\nacc, nl = out['acceptance_probability'], out['n_leapfrog']\n"
                   "assert torch.isfinite(q).all() and torch.isfinite(acc).all() and (nl > 0).all() and acc.mean() > 0.5\n"
                   "print('SNIPPET-OK', float(acc.mean()))\n")
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "SNIPPET-OK" in out.stdout, out.stderr[-3000:]


def test_integration_md_custom_density_snippet_runs(tmp_path):
    """The second snippet of INTEGRATION.md -- a user-defined log-density bound through the C-ABI, differentiated by the
    engine -- executed behind the first one, then one NUTS transition on the new target."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    assert len(blocks) >= 2 and "aehmc_set_custom_target" in blocks[1]
    code = blocks[0].replace('"libaehmc_hip.so"', repr(os.path.join(ROOT, "aehmc_amd", "libaehmc_hip.so")))
    code += blocks[1].replace("/path/to/aehmc_amd/csrc", os.path.join(ROOT, "aehmc_amd", "csrc")).replace("/var/cache/aehmc_rtc", str(tmp_path))
    code += ("\nassert lib.aehmc_new_state(ctx, ct.c_int64(C), ct.c_void_p(q.data_ptr()), ct.c_void_p(U.data_ptr()), ct.c_void_p(g.data_ptr()), stream) == 0\n"
             "torch.cuda.synchronize()\n"
             "ref = (0.5 * 6.0 * torch.log1p(q * q / 5.0)).sum(1)\n"
             "assert torch.allclose(U, ref, rtol=1e-12), (U[:3], ref[:3])\n"
             "assert torch.allclose(g, 6.0 * q / (5.0 + q * q), rtol=1e-12, atol=1e-14)      # the engine's derivative of the density\n"
             "rc = lib.aehmc_nuts_step(ctx, ct.c_int64(C), ct.c_void_p(rng.data_ptr()), ct.c_double(0.1), ct.c_int64(10), ct.c_double(1000.0),\n"
             "                         ct.c_void_p(q.data_ptr()), ct.c_void_p(U.data_ptr()), ct.c_void_p(g.data_ptr()), ct.byref(diag), stream)\n"
             "assert rc == 0, lib.aehmc_last_error(ctx)\n"
             "torch.cuda.synchronize()\n"
             "assert torch.isfinite(q).all() and (out['n_leapfrog'] > 0).all()\n"
             "import os\nassert any(f.endswith('.aehmcco') for f in os.listdir(%r))\n"
             "print('SNIPPET2-OK')\n" % str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "SNIPPET2-OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
