"""SURVEY.md section 5 (sanitizers): the C restatement built with AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C oracle asan`) runs the reference-pinned cases G1 (README.md:22-54) and G2
(examples/LinearRegression.ipynb:293-297), a dense-metric NUTS transition and a many-chain OpenMP call without a
report.  GPU sanitizers are not available on this pool; the HIP side is covered by parity instead."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import c_oracle as co
# G1
t, m = co.Target(co.T_STD_NORMAL, 1), co.Metric(np.float64(1.0), 1)
rng = co.site_states([0], 4)
q, U, g = co.new_state(t, [[0.0]])
r = co.nuts_step(t, m, rng, 1e-2, q, U, g)
assert q[0, 0] == 1.1034719409361107 and r["n_leapfrog"][0] == 136
# G2
r0 = np.random.default_rng(0)
X = r0.normal(0, 1, size=(10_000,)); y = 3 * X + r0.normal(0, 1)
t = co.Target(co.T_LINREG, 2, X=X, y=y)
q, U, g = co.new_state(t, [[3.0, np.log(0.21)]])
r = co.hmc_step(t, co.Metric(np.array([1.0, 1.0]), 2), co.site_states([0], 2), 5e-5, 1024, q, U, g)
np.testing.assert_allclose(q[0], [2.99946192, -1.30494977], atol=5e-9)
# dense metric + dense target, several chains on several threads, deep trees and a cut one
rr = np.random.default_rng(3)
D, C = 17, 12
A, B = rr.normal(size=(D, D)), rr.normal(size=(D, D))
P, imm = A @ A.T / D + np.eye(D), B @ B.T / D + np.eye(D)
t = co.Target(co.T_DENSE_MVN, D, mu=rr.normal(size=D), prec=0.5 * (P + P.T))
for max_exp in (10, 2):
    q, U, g = co.new_state(t, rr.normal(size=(C, D)))
    res = co.nuts_step(t, co.Metric(0.5 * (imm + imm.T), D), co.site_states(list(range(C)), 4), 0.11, q, U, g,
                       max_exp=max_exp, nthreads=4)
    assert np.isfinite(q).all() and (res["n_leapfrog"] > 0).all()
print("SANITIZED-OK")
"""


def test_c_restatement_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan on this host")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "libaehmc_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, AEHMC_ORACLE_LIB=so, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", PROG, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "SANITIZED-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
