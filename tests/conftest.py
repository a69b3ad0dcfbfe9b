import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def regression_data():
    """Data of examples/LinearRegression.ipynb:68-70 (reference notebook, cell 4)."""
    import numpy as np
    rng = np.random.default_rng(0)
    X = rng.normal(0, 1, size=(10_000,))
    y = 3 * X + rng.normal(0, 1)
    return X, y
