"""A FIXED list of cases of the randomised parity sweeps (tools/fuzz_parity.py, tools/fuzz_custom.py): random sampler /
target / metric / sizes / engine options / per-chain parameters / sample() vs calls, against the C oracle.  The case ids
are deterministic (FUZZ_COUNT: the first N ids of a seed), so the test asserts parity only -- never how many
configurations a box gets through in some number of seconds (a cold hipRTC cache compiles each user target: 1.5 s).
The long wall-clock sweeps are tools/ runs recorded under profiles/."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_random_configurations_match_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "0", "7"],
                         capture_output=True, text=True, timeout=550, cwd=ROOT, env=dict(os.environ, FUZZ_COUNT="120"))
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    line = [l for l in out.stdout.splitlines() if l.startswith("fuzz:")][-1]
    assert " 0 mismatches" in line, line


@pytest.mark.timeout(600)
def test_random_user_defined_targets_match_numpy():
    """tools/fuzz_custom.py: user-defined targets (density-only Student-t, a joint AR(1) density, logistic regression) over
    the run-time compiled kernel families, against the numpy restatement with the analytic gradient"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_custom.py"), "0", "11"],
                         capture_output=True, text=True, timeout=550, cwd=ROOT, env=dict(os.environ, FUZZ_COUNT="40"))
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    line = [l for l in out.stdout.splitlines() if l.startswith("fuzz_custom:")][-1]
    assert line.split()[1] == "40" and " 0 mismatches" in line, line
