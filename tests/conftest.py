import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "perf: wall-clock / throughput expectations; selected only by an explicit -m perf")


# Collection order of the GPU suite: what the REFERENCE pins runs first, the newest code last, so that under `-x` a failure
# in a recent feature cannot hide the oldest evidence.  (1) the reference's own values -- README value G1, notebook G2 / G3,
# its unit tables (tests/test_metrics.py:39-120, test_termination.py:12-62, test_trajectory.py:144-208 of the reference),
# device RNG against numpy; (2) committed golden fixtures and the independent dense derivation; (3) BASELINE.json's configs
# at size; (4) one case per compiled kernel family against the oracle; (5) statistics (MCSE) and adaptation; (6) edges and
# boundary; (7) user-defined targets, autodiff; (8) randomised sweeps and the bench.py contract.
_FIRST_IN_PARITY = ("test_g1_readme_bit_exact_on_gpu", "test_g2_g3_regression_on_gpu", "test_kinetic_energy_and_turning_tables",
                    "test_multiplicative_expansion_outcomes_on_gpu", "test_velocity_verlet_analytic",
                    "test_divergent_first_step_keeps_rng_in_step_with_oracle", "test_device_rng_matches_numpy",
                    "test_rng_state_after_many_momentum_draws", "test_config2_full_size_properties_and_subset_parity",
                    "test_config3_full_size_dense_nuts", "test_config5_regression_warmup_properties")
_FILE_ORDER = ("test_gpu_parity.py", "test_golden_fixtures.py", "test_dense_pin.py", "test_gpu_configs.py",
               "test_gpu_block_dense.py", "test_gpu_pc_dense.py", "test_gpu_fp_contract.py", "test_gpu_statistics.py",
               "test_gpu_building_blocks.py", "test_gpu_adaptation.py", "test_gpu_edges.py", "test_gpu_boundary.py",
               "test_gpu_custom_target.py", "test_gpu_autodiff.py", "test_gpu_callable.py", "test_gpu_fuzz.py", "test_gpu_bench_contract.py")


def _rank(item):
    fname = os.path.basename(str(item.fspath))
    if item.get_closest_marker("gpu") is None:
        return (0, 0, 0)  # CPU tests keep their place (stable sort) ahead of the GPU files
    f = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else len(_FILE_ORDER) - 2  # (unknown GPU files: before fuzz / bench)
    base = item.name.split("[")[0]
    first = _FIRST_IN_PARITY.index(base) if (fname == "test_gpu_parity.py" and base in _FIRST_IN_PARITY) else len(_FIRST_IN_PARITY)
    return (1, f, first)


def pytest_collection_modifyitems(config, items):
    if "perf" not in (config.getoption("-m") or ""):  # perf expectations never ride along with `-m gpu` or `-m "not gpu"`
        keep, drop = [], []
        for it in items:
            (drop if it.get_closest_marker("perf") else keep).append(it)
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep
    items.sort(key=_rank)


@pytest.fixture(scope="session")
def regression_data():
    """Data of examples/LinearRegression.ipynb:68-70 (reference notebook, cell 4)."""
    import numpy as np
    rng = np.random.default_rng(0)
    X = rng.normal(0, 1, size=(10_000,))
    y = 3 * X + rng.normal(0, 1)
    return X, y
