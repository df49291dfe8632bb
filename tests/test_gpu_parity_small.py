"""GPU: mcmc::hmc / mcmc::mala / mcmc::rwmh / mcmc::nuts on the d = 2 normal model of the reference's example programs
(examples/eigen/{hmc,mala,nuts}_normal.cpp) vs the CPU oracle through the C ABI -- bit-exact.  One lane per chain
(small_samplers.hpp): any dense precond_mat / cov_mat with any bounds, including bounded MALA with a dense preconditioner."""
import numpy as np
import pytest

import mcmc_amd
import orc

pytestmark = pytest.mark.gpu

M_DENSE = np.array([[1.3, 0.4], [0.4, 0.7]])
M_DIAG = np.diag([0.6, 1.8])
BOUNDS = {
    "lower": (np.array([-np.inf, 0.0]), np.array([np.inf, np.inf])),          # sigma > 0
    "upper": (np.array([-np.inf, -np.inf]), np.array([7.5, 12.0])),
    "box": (np.array([-1.0, 0.1]), np.array([6.0, 9.0])),
    "mixed": (np.array([0.5, -np.inf]), np.array([np.inf, 8.0])),
}


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    return 2.0 + 2.0 * rng.standard_normal(n)


def _init(C, seed=1):
    rng = np.random.default_rng(seed)
    return np.stack([2.0 + rng.uniform(-1.0, 1.0, C), 2.0 + rng.uniform(-0.5, 1.5, C)], axis=1)


CASES = [
    # algo, n_data, C, step, n_leap, burn, keep, precond, bounds
    ("hmc", 1000, 70, 0.08, 1, 10, 40, None, None),             # examples/eigen/hmc_normal.cpp: step 0.08, one leapfrog
    ("hmc", 257, 64, 0.03, 5, 0, 25, None, None),
    ("hmc", 100, 130, 0.05, 3, 4, 20, M_DENSE, None),
    ("hmc", 64, 40, 0.04, 4, 3, 20, None, "lower"),
    ("hmc", 50, 33, 0.04, 2, 2, 24, M_DENSE, "box"),
    ("hmc", 31, 65, 0.05, 3, 0, 16, M_DIAG, "mixed"),
    ("mala", 1000, 70, 0.08, 1, 10, 40, None, None),            # examples/eigen/mala_normal.cpp
    ("mala", 200, 64, 0.10, 1, 0, 30, M_DENSE, None),
    ("mala", 64, 40, 0.08, 1, 3, 30, None, "lower"),
    ("mala", 50, 33, 0.08, 1, 2, 24, M_DENSE, "box"),            # dense precond + bounds: INV(eps^2 J M) per draw
    ("mala", 31, 65, 0.10, 1, 0, 16, M_DIAG, "upper"),
    ("rwmh", 1000, 70, 0.05, 1, 10, 40, None, None),
    ("rwmh", 100, 64, 0.08, 1, 0, 30, M_DENSE, None),
    ("rwmh", 64, 40, 0.10, 1, 3, 30, M_DENSE, "box"),
    ("rwmh", 31, 65, 0.15, 1, 0, 20, None, "mixed"),
]
ALGO = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "rwmh": orc.ALGO_RWMH}


@pytest.mark.parametrize("algo,n,C,step,n_leap,burn,keep,precond,bounds", CASES)
def test_small_samplers_bit_exact_vs_oracle(algo, n, C, step, n_leap, burn, keep, precond, bounds):
    x = _data(n, seed=n)
    init = _init(C, seed=C)
    kw, okw = {}, {}
    if precond is not None:
        kw.update(precond_mat=precond); okw.update(precond=precond)
    if bounds:
        lb, ub = BOUNDS[bounds]
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
        init = np.clip(init, np.where(np.isfinite(lb), lb + 0.3, -np.inf), np.where(np.isfinite(ub), ub - 0.3, np.inf))
    st = mcmc_amd.default_settings(rng_seed_value=17, n_burnin_draws=burn, n_keep_draws=keep, step_size=step, n_leap_steps=n_leap, **kw)
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_NORMAL_MODEL, init, st, y=x, chain0=5)
    t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)
    s = orc.make_settings(seed=17, n_burnin=burn, n_keep=keep, n_leap=n_leap, step=step, W=1, hoist=0, **okw)
    o_draws, o = orc.run_many(ALGO[algo], t, init, s, chain0=5)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert 0 < int(g["n_accept"].sum()) < C * keep


NUTS_CASES = [
    # n_data, C, eps_bar0, n_adapt, max_depth, burn, keep, precond, bounds
    (1000, 70, 1.0, 10, 10, 10, 30, None, None),        # examples/eigen/nuts_normal.cpp shape (defaults), short
    (257, 64, 0.05, 0, 6, 0, 20, None, None),           # no adaptation: the search's step for draw 0, epsilon_bar_0 after it
    (100, 130, 1.0, 8, 5, 8, 16, M_DENSE, None),
    (64, 40, 1.0, 6, 10, 6, 20, None, "lower"),
    (50, 33, 1.0, 12, 4, 4, 20, M_DENSE, "box"),        # adaptation continues into the kept draws
    (31, 65, 1.0, 5, 3, 5, 12, M_DIAG, "mixed"),
    (40, 17, 1.0, 4, 0, 4, 6, None, None),              # max_tree_depth = 0: no tree at all
]


@pytest.mark.parametrize("n,C,eps0,n_adapt,max_depth,burn,keep,precond,bounds", NUTS_CASES)
def test_small_nuts_bit_exact_vs_oracle(n, C, eps0, n_adapt, max_depth, burn, keep, precond, bounds):
    x = _data(n, seed=n)
    init = _init(C, seed=C)
    kw, okw = {}, {}
    if precond is not None:
        kw.update(precond_mat=precond); okw.update(precond=precond)
    if bounds:
        lb, ub = BOUNDS[bounds]
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
        init = np.clip(init, np.where(np.isfinite(lb), lb + 0.3, -np.inf), np.where(np.isfinite(ub), ub - 0.3, np.inf))
    st = mcmc_amd.default_settings(rng_seed_value=23, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps0,
                                   n_adapt_draws=n_adapt, max_tree_depth=max_depth, **kw)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_NORMAL_MODEL, init, st, y=x, chain0=11)
    t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)
    s = orc.make_settings(seed=23, n_burnin=burn, n_keep=keep, step=eps0, n_adapt=n_adapt, max_depth=max_depth, W=1, **okw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=11)
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    if max_depth:
        assert g["depth"].max() <= max_depth and g["depth"].min() >= 1 and len(np.unique(g["depth"])) > 1


def test_small_nuts_tree_depths_match_the_oracle_trace():
    x = _data(120, seed=2)
    init = _init(6, seed=3)
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=6, n_keep_draws=14, n_adapt_draws=6)
    _, g = mcmc_amd.nuts(mcmc_amd.TARGET_NORMAL_MODEL, init, st, y=x)
    t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)
    for c in range(6):
        s = orc.make_settings(seed=4, n_burnin=6, n_keep=14, n_adapt=6, W=1, chain_id=c)
        _, o = orc.run_chain(orc.ALGO_NUTS, t, init[c], s, traces=True)
        assert np.array_equal(g["depth"][:, c], o["depth"])


def test_small_nuts_resume_is_bit_identical_to_one_run():
    x = _data(200, seed=3)
    init = _init(48, seed=9)
    kw = dict(rng_seed_value=5, n_adapt_draws=5)
    full, _ = mcmc_amd.nuts(mcmc_amd.TARGET_NORMAL_MODEL, init, mcmc_amd.default_settings(n_burnin_draws=7, n_keep_draws=20, **kw), y=x)
    a, ia = mcmc_amd.nuts(mcmc_amd.TARGET_NORMAL_MODEL, init, mcmc_amd.default_settings(n_burnin_draws=7, n_keep_draws=8, **kw), y=x)
    b, _ = mcmc_amd.nuts(mcmc_amd.TARGET_NORMAL_MODEL, ia["theta"].T, mcmc_amd.default_settings(n_burnin_draws=0, n_keep_draws=12, **kw),
                         y=x, draw0=15, step_size_in=ia["eps"])
    assert np.array_equal(np.concatenate([a, b]), full)
