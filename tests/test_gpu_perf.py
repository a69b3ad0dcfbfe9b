"""Wall-clock and throughput expectations.  Marker `perf`, NOT `gpu`: the correctness suite (`-m gpu`) holds no assertion
that a correct library can fail on a slower or colder box; these run on demand (`pytest -m perf`, on a GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.perf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_c2_reaches_the_north_star_rate():
    assert _bench("--config", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")["value"] > 1e8


def test_secondary_entries_keep_their_rates():
    d = _bench("--steps", "2", "--no-cpu-baseline")
    by = {e["config"]: e for e in d["secondary"]}
    for k in ("dense-nuts-d100", "dense-nuts-d200"):
        assert by[k]["ms_per_transition"] < 1.5
    assert by["pc-dense-nuts-d200"]["roofline"]["frac"] > 0.3
    assert by["diag-hmc-fp_contract"]["value"] > 1.15 * by["diag-hmc"]["value"]


def test_disk_cache_makes_the_second_process_faster(tmp_path):
    prog = r"""
import json, sys, time, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from aehmc_amd import RandomStream, nuts, targets
from test_gpu_autodiff import FUNNEL
torch.zeros(1, device="cuda")
tgt = targets.CustomJoint(FUNNEL, dim=10)
q0 = torch.as_tensor(0.3 * np.random.default_rng(0).normal(size=(8, 10)), device="cuda")
t0 = time.perf_counter()
state = nuts.new_state(q0, tgt)
info, _ = nuts.new_kernel(RandomStream(seeds=list(range(8))), tgt, max_num_expansions=5)(state, 0.1, np.ones(10))
torch.cuda.synchronize()
print(json.dumps({"seconds": time.perf_counter() - t0}))
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, AEHMC_AMD_RTC_CACHE=str(tmp_path))
    secs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        secs.append(json.loads(out.stdout.strip().splitlines()[-1])["seconds"])
    assert secs[1] < 0.5 * secs[0] and secs[1] < 2.0, secs
