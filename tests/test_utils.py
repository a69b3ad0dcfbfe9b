"""``utils.RaveledParamsMap`` with the cases of the reference's tests (/root/reference/tests/test_utils.py:11-74),
on eager arrays (numpy and torch) instead of symbolic variables."""
import numpy as np
import pytest

from aehmc_amd.utils import RaveledParamsMap


def test_RaveledParamsMap():
    beta_size, kappa_size = (3, 2), (20,)
    ref = {"beta": np.zeros(beta_size), "tau": np.float64(0.0), "kappa": np.zeros(kappa_size)}
    rp_map = RaveledParamsMap(ref)
    assert repr(rp_map) == "RaveledParamsMap((beta, tau, kappa))"
    exp_beta_part = np.exp(np.arange(np.prod(beta_size)).reshape(beta_size))
    exp_tau_part = 1.0
    exp_kappa_part = np.exp(np.arange(np.prod(kappa_size)).reshape(kappa_size))
    exp_raveled = np.concatenate([exp_beta_part.ravel(), np.atleast_1d(exp_tau_part), exp_kappa_part.ravel()])
    assert np.array_equal(rp_map.ravel_params([exp_beta_part, exp_tau_part, exp_kappa_part]), exp_raveled)
    assert np.array_equal(rp_map.ravel_params({"kappa": exp_kappa_part, "beta": exp_beta_part, "tau": exp_tau_part}), exp_raveled)
    parts = rp_map.unravel_params(exp_raveled)
    assert np.array_equal(parts["beta"], exp_beta_part) and parts["beta"].shape == beta_size
    assert np.array_equal(parts["tau"], exp_tau_part) and parts["tau"].shape == ()
    assert np.array_equal(parts["kappa"], exp_kappa_part)
    with pytest.raises(ValueError):
        rp_map.unravel_params(exp_raveled[:-1])


def test_RaveledParamsMap_dtype():
    rp_map = RaveledParamsMap({"tau": np.float64(0.3), "lmbda": np.int64(4)})
    q = rp_map.ravel_params((np.float64(0.3), np.int64(4)))
    parts = rp_map.unravel_params(q)
    assert parts["tau"].dtype == np.float64 and parts["lmbda"].dtype == np.int64 and parts["lmbda"] == 4


def test_RaveledParamsMap_chain_axis_and_torch():
    """a leading chain axis ([C, ...] blocks <-> the [C, D] position the kernels take), torch tensors in and out"""
    torch = pytest.importorskip("torch")
    C = 5
    ref = {"w": torch.zeros(C, 3, 2, dtype=torch.float64), "b": torch.zeros(C, dtype=torch.float64)}
    m = RaveledParamsMap(ref, batch_ndim=1)
    assert m.size == 7
    w = torch.arange(C * 6, dtype=torch.float64).reshape(C, 3, 2)
    b = -torch.arange(C, dtype=torch.float64)
    q = m.ravel_params([w, b])
    assert q.shape == (C, 7) and torch.equal(q[:, :6], w.reshape(C, 6)) and torch.equal(q[:, 6], b)
    parts = m.unravel_params(q)
    assert torch.equal(parts["w"], w) and torch.equal(parts["b"], b)
    hist = torch.stack([q, 2 * q])  # [N, C, D] samples unravel with their leading axes
    assert m.unravel_params(hist)["w"].shape == (2, C, 3, 2)
