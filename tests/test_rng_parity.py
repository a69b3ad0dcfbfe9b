"""The restated RNG (PCG64 + numpy ziggurat normal + binomial(1,p) inversion, scheme A)
must reproduce numpy.random.Generator bit for bit -- it is what the reference's
RandomStream call sites execute."""
import numpy as np
import pytest

from oracle import c_oracle as co


@pytest.mark.parametrize("seed", [0, 1, 59, 2**40 + 7])
def test_c_rng_matches_numpy(seed):
    st = co.site_states([seed], 4)
    children = np.random.SeedSequence(seed).spawn(4)
    g = np.random.default_rng(children[0])
    n = 400_000  # ~100 tail draws and ~3000 wedge rejections
    assert np.array_equal(co.rng_normals(st[0, 0], n), g.normal(0, 1, size=n))
    g2 = np.random.default_rng(children[1])
    ps = np.random.default_rng(7).random(50_000)
    ps[::7], ps[::11], ps[::13] = 0.0, 1.0, 0.5
    assert np.array_equal(co.rng_bernoulli(st[0, 1], ps),
                          np.array([g2.binomial(1, p) for p in ps]))
    # both generators left in the same state (p == 0 draws nothing)
    assert co.rng_doubles(st[0, 1], 8).tolist() == g2.random(8).tolist()
    assert co.rng_doubles(st[0, 0], 8).tolist() == g.random(8).tolist()


def test_site_order_is_spawn_order():
    st = co.site_states([5, 6], 3, first_site=1)
    ch = np.random.SeedSequence(6).spawn(4)
    s = np.random.PCG64(ch[2]).state["state"]
    assert int(st[1, 1, 0]) == s["state"] >> 64 and int(st[1, 1, 3]) == s["inc"] & (2**64 - 1)
