"""CPU: the ESS estimator behind bench.py's ESS/sec (the reference has no ESS code)."""
import numpy as np

from mcmc_amd.ess import ess_min_total, ess_per_chain


def test_iid_draws_have_ess_close_to_n():
    x = np.random.default_rng(0).standard_normal((200, 3, 512))
    e = ess_per_chain(x)
    assert np.all(np.abs(e / 200 - 1) < 0.1)


def test_ar1_matches_theory():
    rng = np.random.default_rng(1)
    n, d, C, phi = 400, 2, 1024, 0.8
    y = np.zeros((n, d, C))
    y[0] = rng.standard_normal((d, C))
    for t in range(1, n):
        y[t] = phi * y[t - 1] + np.sqrt(1 - phi ** 2) * rng.standard_normal((d, C))
    want = n * (1 - phi) / (1 + phi)
    e = ess_per_chain(y)
    assert np.all(np.abs(e / want - 1) < 0.15)
    tot, _ = ess_min_total(y, n_chains_total=10 * C)
    assert abs(tot / (10 * C * want) - 1) < 0.2


def test_short_runs_do_not_crash():
    assert ess_per_chain(np.zeros((2, 3, 4))).shape == (3,)
