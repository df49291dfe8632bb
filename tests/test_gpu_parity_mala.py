"""GPU: MALA vs the CPU oracle through the C ABI -- bit-exact."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # kind,   d,   C,  eps,  burn, keep
    ("dense", 8, 16, 0.30, 5, 20),      # SURVEY 8(c) golden shape
    ("iso", 3, 7, 0.50, 10, 30),        # 3-D isotropic Gaussian
    ("dense", 128, 64, 0.08, 4, 10),    # d=128 correlated Gaussian
    ("dense", 100, 130, 0.05, 2, 6),    # ragged d and C
    ("diag", 40, 48, 0.05, 3, 9),
    ("dense", 64, 256, 0.40, 0, 12),    # large step: many rejections
]


@pytest.mark.parametrize("kind,d,C,eps,burn,keep", CASES)
def test_mala_bit_exact_vs_oracle(kind, d, C, eps, burn, keep):
    init = synth.initial_states(C, d, seed=31)
    prec, k_gpu, k_orc = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    if kind == "dense":
        prec, k_gpu, k_orc = synth.dense_gaussian_precision(d, seed=7), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
    elif kind == "diag":
        prec, k_gpu, k_orc = synth.ill_conditioned_diag(d, 30.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    st = mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    g_draws, g = mcmc_amd.mala(k_gpu, init, st, prec=prec, chain0=77)
    t = orc.TargetSpec(k_orc, d, prec=prec, W=4)
    s = orc.make_settings(seed=99, n_burnin=burn, n_keep=keep, step=eps, W=4, hoist=1)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=77)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.linalg.norm(g_draws - o_draws) <= 1e-9 * np.linalg.norm(o_draws)
    assert np.array_equal(g["theta"], o_draws[-1])


def test_mala_rejects_and_samples_the_target():
    d, C = 16, 4096
    prec = synth.dense_gaussian_precision(d, seed=7)
    init = synth.initial_states(C, d, seed=1)
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=300, n_keep_draws=50, step_size=0.35)
    draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    acc = g["n_accept"].mean() / 50
    assert 0.3 < acc < 0.99
    last = draws[-1]
    cov = last @ last.T / C
    assert abs(np.trace(cov) / np.trace(np.linalg.inv(prec)) - 1) < 0.1


# ---------------------------------------------------------------- logistic-regression target (BASELINE config 3)
LOGIT_CASES = [
    # d,   N,    C,  eps,   burn, keep
    (5, 40, 16, 0.10, 5, 20),        # SURVEY 8(c) golden shape "logistic d=5"
    (64, 100, 20, 0.05, 3, 8),       # one tile per wave, ragged N and C
    (100, 333, 40, 0.03, 2, 6),      # d_pad 128, ragged everything
    (512, 1024, 32, 0.02, 2, 4),     # config 3 dimensions
    (300, 64, 17, 0.20, 0, 10),      # big step: rejections
]


def _blocks(d):
    dq = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    return 4, dq


@pytest.mark.parametrize("d,N,C,eps,burn,keep", LOGIT_CASES)
def test_mala_logistic_bit_exact_vs_oracle(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    nb, bs = _blocks(d)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=nb, block_size=bs, eta_chains=2)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=eps, W=4, hoist=1, blocks=nb, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=3)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.linalg.norm(g_draws - o_draws) <= 1e-9 * np.linalg.norm(o_draws)


def test_mala_logistic_posterior_is_sane():
    d, N, C = 8, 400, 2048
    X, y = synth.logistic_problem(d, N, seed=9)
    init = np.zeros((C, d))
    st = mcmc_amd.default_settings(rng_seed_value=2, n_burnin_draws=300, n_keep_draws=20, step_size=0.25)
    draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    assert 0.3 < g["n_accept"].mean() / 20 < 0.999
    post_mean = draws[-1].mean(axis=1)
    # the posterior mean is close to the mode: gradient of log K at it is near zero relative to the prior scale
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y)
    _, grad = t.kernel(post_mean)
    assert np.abs(grad).max() < 2.0


# ---------------------------------------------------------------- box constraints and diagonal precond_mat
def _bounds(d, seed=0):
    """A mix of the four bounds types of determine_bounds_type.hpp:27-57."""
    rng = np.random.default_rng(seed)
    kind = rng.integers(1, 5, d)
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf)
    ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    return lb, ub


@pytest.mark.parametrize("d,C,eps,burn,keep", [(6, 16, 0.15, 4, 12), (128, 48, 0.03, 2, 5), (37, 20, 0.08, 0, 6)])
def test_bounded_mala_bit_exact_vs_oracle(d, C, eps, burn, keep):
    lb, ub = _bounds(d, seed=d)
    prec = synth.dense_gaussian_precision(d, seed=5)
    init = np.clip(synth.initial_states(C, d, seed=14) * 0.3, -1.0, 1.5)     # inside every box
    st = mcmc_amd.default_settings(rng_seed_value=78, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps,
                                   vals_bound=1, lower_bounds=lb, upper_bounds=ub)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=78, n_burnin=burn, n_keep=keep, step=eps, W=4, lower=lb, upper=ub)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=4)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert ((g_draws >= lb[None, :, None]) & (g_draws <= ub[None, :, None])).all()   # reported in the constrained space
    assert 0 < g["n_accept"].sum() < C * keep                                          # both branches of the accept step ran


@pytest.mark.parametrize("d,C,eps,bounded", [(8, 16, 0.3, False), (128, 40, 0.05, False), (20, 24, 0.15, True)])
def test_diagonal_precond_mala_bit_exact_vs_oracle(d, C, eps, bounded):
    prec = synth.dense_gaussian_precision(d, seed=5)
    M = np.diag(1.0 / np.diag(prec) * np.linspace(0.5, 2.0, d))
    init = np.clip(synth.initial_states(C, d, seed=15) * 0.3, -1.0, 1.5)
    kw, okw = {}, {}
    if bounded:
        lb, ub = _bounds(d, seed=3)
        kw = dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub)
        okw = dict(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=3, n_keep_draws=7, step_size=eps, precond_mat=M, **kw)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=6, n_burnin=3, n_keep=7, step=eps, W=4, precond=M, **okw)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.5, 1.5, d))


@pytest.mark.parametrize("d,C,eps", [(8, 16, 0.3), (37, 40, 0.1), (64, 33, 0.08), (128, 40, 0.05), (90, 17, 0.06)])   # d > 64: M, L, INV(Sigma) from L2
def test_dense_precond_mala_bit_exact_vs_oracle(d, C, eps):
    prec = synth.dense_gaussian_precision(d, seed=5)
    M = _spd(d, seed=d)
    init = synth.initial_states(C, d, seed=15) * 0.3
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=3, n_keep_draws=7, step_size=eps, precond_mat=M)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=1)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=6, n_burnin=3, n_keep=7, step=eps, W=4, precond=M, hoist=1)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=1)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert 0 < g["n_accept"].sum()


def test_dense_precond_mala_with_bounds_runs_literally():
    """ref: src/mala.cpp:152-157, include/mcmc/mala.ipp:45-55 -- INV(eps^2 J(theta') M) per draw is O(d^3) in the reference too: the
    configuration runs on the literal kernel (mcmc_amd/csrc/literal.hpp), every chain, not on an MFMA kernel (round 2 refused it)"""
    d, C = 8, 20
    M = _spd(d, seed=2)
    lb, ub = np.full(d, -1.0), np.full(d, 1.0)
    init = synth.initial_states(C, d, seed=3) * 0.2
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=2, n_keep_draws=6, step_size=0.2, precond_mat=M, vals_bound=1,
                                   lower_bounds=lb, upper_bounds=ub)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_ISO, init, st)
    s = orc.make_settings(seed=4, n_burnin=2, n_keep=6, step=0.2, W=4, precond=M, hoist=1, lower=lb, upper=ub)
    o_draws, o = orc.run_many(orc.ALGO_MALA, orc.TargetSpec(orc.TARGET_ISO, d, W=4), init, s)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert mcmc_amd.last_kernel() == "literal_kernel<1>"
