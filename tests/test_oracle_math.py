"""CPU: the oracle's RNG / deterministic math against published KATs and libm (not gpu)."""
import numpy as np

import orc


def test_philox4x32_10_random123_kats():
    # Random123 kat_vectors (Salmon et al., SC'11): philox4x32 10 rounds
    assert [hex(v) for v in orc.philox([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in orc.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                       [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def _ulps(a, ref):
    return np.abs(a - ref) / np.spacing(np.abs(ref))


def test_exp_log_accuracy_vs_libm():
    x = np.linspace(-700, 700, 100001)
    assert _ulps(orc.math_eval(0, x)[0], np.exp(x)).max() <= 4
    x = np.exp(np.linspace(-700, 700, 100001))
    assert _ulps(orc.math_eval(1, x)[0], np.log(x)).max() <= 6
    x = np.concatenate([np.linspace(0.5, 2, 50001), 1 + np.logspace(-15, -1, 1000)])
    x = x[x != 1.0]
    assert _ulps(orc.math_eval(1, x)[0], np.log(x)).max() <= 6


def test_exp_log_special_values():
    e, _ = orc.math_eval(0, np.array([0.0, 710.0, -800.0, np.inf, -np.inf, np.nan, 0.01]))
    assert e[0] == 1.0 and e[1] == np.inf and e[2] == 0.0 and e[3] == np.inf and e[4] == 0.0 and np.isnan(e[5])
    assert abs(e[6] - np.exp(0.01)) < 1e-15
    l, _ = orc.math_eval(1, np.array([1.0, 0.0, -1.0, np.inf, 5e-324, 2.0, 0.5]))
    assert l[0] == 0.0 and l[1] == -np.inf and np.isnan(l[2]) and l[3] == np.inf
    assert abs(l[4] - np.log(5e-324)) < 1e-12
    assert l[5] == -l[6]


def test_sincos2pi_accuracy():
    u = np.random.default_rng(0).random(100001)
    s, c = orc.math_eval(2, u)
    assert np.abs(s - np.sin(2 * np.pi * u)).max() < 2e-15
    assert np.abs(c - np.cos(2 * np.pi * u)).max() < 2e-15
    assert np.abs(s * s + c * c - 1).max() < 1e-15
    s, c = orc.math_eval(2, np.array([0.0, 0.125, 0.25, 0.5, 0.75]))
    np.testing.assert_allclose(s, [0, np.sqrt(0.5), 1, 0, -1], atol=2e-16)
    np.testing.assert_allclose(c, [1, np.sqrt(0.5), 0, -1, 0], atol=2e-16)


def test_uniform_open_interval_and_moments():
    u = np.array([orc.uniform(7, c, 3, 0) for c in range(20000)])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # exactness of the (2k+1) 2^-53 grid
    assert np.all((u * 2.0 ** 53) % 2 == 1)


def test_normals_moments_and_slot_map():
    z = orc.normal_vec(1, 2, 3, 0, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3) < 0.08
    # prefix property: dimension i's normal does not depend on d (ragged d shares a stream)
    z10, z128 = orc.normal_vec(5, 9, 1, 0, 10), orc.normal_vec(5, 9, 1, 0, 128)
    assert np.array_equal(z10, z128[:10])
    # distinct chains / draws / streams decorrelate
    assert not np.array_equal(orc.normal_vec(5, 9, 1, 0, 8), orc.normal_vec(5, 10, 1, 0, 8))
    assert not np.array_equal(orc.normal_vec(5, 9, 1, 0, 8), orc.normal_vec(5, 9, 2, 0, 8))
    assert not np.array_equal(orc.normal_vec(5, 9, 1, 0, 8), orc.normal_vec(5, 9, 1, 2, 8))


def test_dot_reduction_orders():
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(37), rng.standard_normal(37)
    exact = float(np.dot(x, y))
    for W in (1, 4, 64):
        assert abs(orc.dot(x, y, W) - exact) < 1e-13
    # W=4: four strided fma chains combined as (q0+q2)+(q1+q3)
    import math
    q = [0.0] * 4
    for i in range(37):
        q[i % 4] = math.fma(x[i], y[i], q[i % 4]) if hasattr(math, "fma") else None
    if q[0] is not None:
        assert orc.dot(x, y, 4) == (q[0] + q[2]) + (q[1] + q[3])
