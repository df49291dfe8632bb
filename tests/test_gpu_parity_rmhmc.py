"""GPU: mcmc::rmhmc (many chains) vs the CPU oracle through the C ABI -- bit-exact.  SURVEY 8 (f-4).
Target: the d = 2 normal model of the reference's example (examples/eigen/rmhmc_normal.cpp) with its Fisher metric."""
import numpy as np
import pytest

import mcmc_amd
import orc

pytestmark = pytest.mark.gpu


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    return 2.0 + 2.0 * rng.standard_normal(n)


def _init(C, seed=1):
    rng = np.random.default_rng(seed)
    return np.stack([2.0 + rng.uniform(-1.0, 1.0, C), 2.0 + rng.uniform(-0.5, 1.5, C)], axis=1)


CASES = [
    # n_data, C, eps, n_leap, n_fp, burn, keep, bounds
    (1000, 70, 0.02, 1, 5, 10, 40, None),            # the example's shape (step reduced: see DESIGN, sign of the reference)
    (1000, 33, 0.20, 1, 5, 5, 30, None),             # the example's own step size
    (257, 64, 0.05, 3, 2, 0, 25, None),
    (100, 130, 0.05, 2, 5, 4, 20, "lower"),           # sigma > 0
    (64, 40, 0.03, 2, 3, 3, 20, "box"),               # mu in (-1, 6), sigma in (0.1, 9)
    (50, 17, 0.05, 1, 0, 2, 12, None),                # no fixed-point iterations
    (31, 65, 0.04, 4, 1, 0, 10, "upper"),
]


def _bounds(which):
    if which == "lower":
        return np.array([-np.inf, 0.0]), np.array([np.inf, np.inf])
    if which == "upper":
        return np.array([-np.inf, -np.inf]), np.array([7.5, 12.0])
    return np.array([-1.0, 0.1]), np.array([6.0, 9.0])


@pytest.mark.parametrize("n,C,eps,n_leap,n_fp,burn,keep,bounds", CASES)
def test_rmhmc_bit_exact_vs_oracle(n, C, eps, n_leap, n_fp, burn, keep, bounds):
    x = _data(n, seed=n)
    init = _init(C, seed=C)
    kw, okw = {}, {}
    if bounds:
        lb, ub = _bounds(bounds)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=41, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps,
                                   n_leap_steps=n_leap, n_fp_steps=n_fp, **kw)
    g_draws, g = mcmc_amd.rmhmc(mcmc_amd.TARGET_NORMAL_MODEL, init, st, y=x, chain0=123)
    t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)
    s = orc.make_settings(seed=41, n_burnin=burn, n_keep=keep, n_leap=n_leap, step=eps, n_fp=n_fp, W=1, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RMHMC, t, init, s, chain0=123)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g_draws, o_draws)
    assert 0 < int(g["n_accept"].sum()) < C * keep        # both branches of the accept step were taken


def test_rmhmc_resume_is_bit_identical_to_one_run():
    x = _data(200, seed=3)
    init = _init(48, seed=9)
    kw = dict(rng_seed_value=5, step_size=0.03, n_leap_steps=2, n_fp_steps=4)
    full, _ = mcmc_amd.rmhmc(mcmc_amd.TARGET_NORMAL_MODEL, init, mcmc_amd.default_settings(n_burnin_draws=6, n_keep_draws=20, **kw), y=x)
    a, ia = mcmc_amd.rmhmc(mcmc_amd.TARGET_NORMAL_MODEL, init, mcmc_amd.default_settings(n_burnin_draws=6, n_keep_draws=8, **kw), y=x)
    b, _ = mcmc_amd.rmhmc(mcmc_amd.TARGET_NORMAL_MODEL, ia["theta"].T, mcmc_amd.default_settings(n_burnin_draws=0, n_keep_draws=12, **kw),
                          y=x, draw0=14)
    assert np.array_equal(np.concatenate([a, b]), full)


def test_rmhmc_rejects_what_it_does_not_implement():
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1)
    with pytest.raises(mcmc_amd.MiMcmcError) as e:        # two d x d x d cubes per chain: d <= 64 on the literal kernel
        mcmc_amd.rmhmc(mcmc_amd.TARGET_GAUSS_ISO, np.zeros((4, 70)), st)
    assert e.value.code == mcmc_amd.MI_ERR_UNSUPPORTED and "d x d x d" in str(e.value)


@pytest.mark.parametrize("kind,d,bounded", [("dense", 6, False), ("iso", 5, True), ("diag", 9, False), ("dense", 20, False)])
def test_rmhmc_gaussian_targets_with_their_constant_metric_run_on_the_literal_kernel(kind, d, bounded):
    """ref: src/rmhmc.cpp:30-287 with G = the precision, dG = 0 (the oracle's orc_target_tensor for the Gaussian kinds): round 2 had no
    metric for them on the device path; literal.hpp runs the sampler as written, one workgroup per chain"""
    from mcmc_amd import synth
    C = 9
    prec = synth.dense_gaussian_precision(d, seed=3) if kind == "dense" else synth.ill_conditioned_diag(d, 30.0) if kind == "diag" else None
    kg, ko = {"dense": (mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE), "diag": (mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG),
              "iso": (mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)}[kind]
    init = synth.initial_states(C, d, seed=7) * 0.4
    kw, okw = {}, {}
    if bounded:
        lb, ub = np.full(d, -2.0), np.full(d, 2.5)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=11, n_burnin_draws=2, n_keep_draws=6, step_size=0.1, n_leap_steps=2, n_fp_steps=3, **kw)
    g_draws, g = mcmc_amd.rmhmc(kg, init, st, prec=prec, chain0=2)
    assert mcmc_amd.last_kernel() == "literal_kernel<4>"
    s = orc.make_settings(seed=11, n_burnin=2, n_keep=6, n_leap=2, step=0.1, n_fp=3, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RMHMC, orc.TargetSpec(ko, d, prec=prec, W=4), init, s, chain0=2)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws) and np.array_equal(g["n_leap"], o["n_leap"])


# ---------------------------------------------------------------- beyond d = 2: Bayesian logistic regression, Fisher metric
@pytest.mark.parametrize("d,N,C,eps,n_leap,n_fp,burn,keep,bounded", [(3, 60, 70, 0.05, 2, 5, 4, 16, False), (4, 80, 33, 0.04, 3, 3, 3, 12, False),
                                                                  (1, 30, 64, 0.08, 2, 4, 2, 12, False), (2, 40, 40, 0.05, 2, 5, 3, 12, True),
                                                                  (4, 50, 17, 0.03, 1, 0, 0, 10, True)])
def test_rmhmc_logistic_fisher_metric_bit_exact_vs_oracle(d, N, C, eps, n_leap, n_fp, burn, keep, bounded):
    """mcmc::rmhmc with a position-dependent d x d metric, d up to 4 (per lane: the d x d inverse and d products Ginv dG_i per
    fixed-point step, ref: src/rmhmc.cpp:99-150,199-272): G = X^T diag(s (1 - s)) X + I on MI_TARGET_LOGISTIC."""
    from mcmc_amd import synth
    X, y = synth.logistic_problem(d, N, seed=8)
    init = synth.initial_states(C, d, seed=7) * 0.3
    kw, okw = {}, {}
    if bounded:
        lb, ub = np.full(d, -2.0), np.full(d, 2.5)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=43, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps, n_leap_steps=n_leap,
                                   n_fp_steps=n_fp, **kw)
    g_draws, g = mcmc_amd.rmhmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=5)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=16, eta_chains=2)
    s = orc.make_settings(seed=43, n_burnin=burn, n_keep=keep, n_leap=n_leap, step=eps, n_fp=n_fp, W=4, blocks=4, block_size=16, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RMHMC, t, init, s, chain0=5)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert 0 < int(g["n_accept"].sum())


@pytest.mark.parametrize("d,N,bounded", [(5, 30, False), (12, 40, False), (7, 25, True)])
def test_rmhmc_logistic_beyond_4_dims_runs_on_the_literal_kernel(d, N, bounded):
    """the Fisher metric X' Lambda X + I with its d x d x d derivative beyond the one-lane engine's d <= 4"""
    from mcmc_amd import synth
    C = 7
    X, y = synth.logistic_problem(d, N, seed=8)
    init = synth.initial_states(C, d, seed=7) * 0.3
    kw, okw = {}, {}
    if bounded:
        lb, ub = np.full(d, -2.0), np.full(d, 2.5)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=43, n_burnin_draws=2, n_keep_draws=6, step_size=0.05, n_leap_steps=2, n_fp_steps=3, **kw)
    g_draws, g = mcmc_amd.rmhmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=5)
    assert mcmc_amd.last_kernel() == "literal_kernel<4>"
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=16, eta_chains=2)
    s = orc.make_settings(seed=43, n_burnin=2, n_keep=6, n_leap=2, step=0.05, n_fp=3, W=4, blocks=4, block_size=16, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RMHMC, t, init, s, chain0=5)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert 0 < int(g["n_accept"].sum())
