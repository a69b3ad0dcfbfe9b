"""Known-answer tests of the reference's warm-up building blocks against the numpy
restatement (oracle/np_adaptation.py): tests/test_adaptation.py:9-22,
tests/test_algorithms.py:10-133, tests/test_mass_matrix.py (reference paths)."""
import numpy as np
import pytest

from oracle import np_adaptation as na


@pytest.mark.parametrize("num_steps, expected", [
    (19, [(0, False)] * 19),
    (100, [(0, False)] * 15 + [(1, False)] * 74 + [(1, True)] + [(0, False)] * 10),
    (200, [(0, False)] * 75 + [(1, False)] * 24 + [(1, True)] + [(1, False)] * 49 + [(1, True)]
     + [(0, False)] * 50),
])
def test_schedule_tables(num_steps, expected):
    s = na.build_schedule(num_steps)
    assert len(s) == num_steps and s == expected


@pytest.mark.parametrize("num_steps, expected", [
    (19, [(0, False)] * 19),
    (100, [(0, False)] * 15 + [(1, False)] * 74 + [(1, True)] + [(0, False)] * 10),
    (200, [(0, False)] * 75 + [(1, False)] * 24 + [(1, True)] + [(1, False)] * 49 + [(1, True)]
     + [(0, False)] * 50),
])
def test_product_schedule_tables(num_steps, expected):
    """The PRODUCT's build_schedule (aehmc_amd/window_adaptation.py, host logic, importable without
    a GPU) against the reference's own tables, tests/test_adaptation.py:9-22."""
    from aehmc_amd.window_adaptation import build_schedule
    s = build_schedule(num_steps)
    assert len(s) == num_steps and s == expected


def test_product_schedule_equals_restatement():
    """... and against the literal restatement of window_adaptation.py:230-327 for every length
    (incl. non-default buffer sizes)."""
    from aehmc_amd.window_adaptation import build_schedule
    for n in range(0, 3000):
        assert build_schedule(n) == na.build_schedule(n), n
    for n in (150, 500, 1234):
        for kw in (dict(initial_buffer_size=20, final_buffer_size=30, first_window_size=10),
                   dict(initial_buffer_size=100, final_buffer_size=10, first_window_size=40)):
            assert build_schedule(n, **kw) == na.build_schedule(n, **kw), (n, kw)


def test_schedule_1000_windows():
    s = na.build_schedule(1000)
    ends = [i for i, (_, e) in enumerate(s) if e]
    assert ends == [99, 149, 249, 449, 949]  # Stan-style doubling windows 25,50,100,200,500


def test_dual_averaging_finds_minimum():
    # tests/test_algorithms.py:10-54: minimise (x-1)^2 with gamma=0.5, mu=0.5, 100 steps
    init, update = na.dual_averaging(gamma=0.5)
    st = init(0.5)
    for _ in range(100):
        st = update(2 * (st.iterates - 1), st)
    assert st.iterates == pytest.approx(1.0, rel=1e-2)
    assert st.iterates_avg == pytest.approx(1.0, rel=1e-2)


@pytest.mark.parametrize("n_dim", [0, 1, 3])
@pytest.mark.parametrize("full", [True, False])
def test_welford_tables(n_dim, full):
    init, update, final = na.welford_covariance(full)
    st = init(n_dim)
    for i in range(10):
        st = update(i * np.ones(n_dim) if n_dim else np.float64(i), *st)
    np.testing.assert_allclose(st[0], 4.5 * (np.ones(n_dim) if n_dim else 1.0))
    cov = final(st[1], st[2])
    expected = 55.0 / 6.0 * (np.ones((n_dim, n_dim)) if (full and n_dim) else
                             (np.ones(n_dim) if n_dim else 1.0))
    assert np.shape(cov) == np.shape(expected)
    np.testing.assert_allclose(cov, expected)
    # constant samples -> zero variance (tests/test_algorithms.py:57-90)
    st = init(n_dim)
    for _ in range(10):
        st = update(np.ones(n_dim) if n_dim else np.float64(1.0), *st)
    np.testing.assert_allclose(final(st[1], st[2]), 0 * expected)


@pytest.mark.parametrize("full", [True, False])
def test_covariance_adaptation_recovers_target(full):
    # tests/test_mass_matrix.py:11-60: 2000 MVN draws -> imm ~ covariance (rtol 0.1)
    r = np.random.default_rng(0)
    cov = np.array([[1.0, 0.3], [0.3, 2.0]])
    xs = r.multivariate_normal([0.0, 3.0], cov, size=2000)
    init, update, final = na.covariance_adaptation(full)
    imm, st = init(2)
    assert imm.shape == ((2, 2) if full else (2,))
    for x in xs:
        st = update(x, st)
    est = final(st)
    np.testing.assert_allclose(est, cov if full else np.diag(cov), rtol=0.1, atol=0.03)


def test_window_adaptation_quirks():
    """first step size is exp(0) = 1 whatever initial_step_size is; mu is
    the step size itself; the last step returns exp(x_avg)."""
    init, update = na.window_adaptation(30, initial_step_size=0.37)
    (da, mm), (eps, imm) = init(np.zeros(2))
    assert eps == 1.0 and da.shrinkage_pts == 0.37 and np.array_equal(imm, np.ones(2))
    ws, params = (da, mm), (eps, imm)
    for step in range(30):
        ws, params = update(step, ws, params, np.ones(2) * step, 0.5)
    assert params[0] == pytest.approx(np.exp(ws[0].iterates_avg))
