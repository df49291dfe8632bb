"""GPU: dense Gaussian targets BEYOND d = 512 (ref: src/hmc.cpp:155-205, src/mala.cpp:149-186, src/rwmh.cpp:123-151 -- n_vals is unrestricted;
VERDICT r5 missing #5).  The state lives in HBM and every gradient evaluation of all chains is one fp64 matrix product W = P Theta on the matrix
cores with the leapfrog's half-kicks and drift in its epilogue (mcmc_amd/csrc/gemm_samplers.hip).  Bit for bit against the oracle (rows of P theta as one
ascending fma chain, dot products as four strided chains -- the orders of the literal kernel that served these shapes before), against that literal
kernel of the same library on more chains, across a continuation, and in the non-finite regime (flagged chains are replayed literally)."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu
ALGO = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "rwmh": orc.ALGO_RWMH}
EPS = {"hmc": 0.02, "mala": 0.03, "rwmh": 0.012}


def _settings(algo, seed, burn, keep, L, eps):
    return mcmc_amd.default_settings(rng_seed_value=seed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)


def _oracle(algo, d, prec, init, seed, burn, keep, L, eps, chain0=0):
    s = orc.make_settings(seed=seed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1)
    return orc.run_many(ALGO[algo], orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s, chain0=chain0)


@pytest.mark.parametrize("algo", ["hmc", "mala", "rwmh"])
@pytest.mark.parametrize("d,C,L", [(513, 45, 3), (640, 130, 1), (1000, 20, 4), (1024, 45, 2), (1100, 7, 3)])
def test_matrix_product_samplers_equal_the_oracle(algo, d, C, L):
    """ragged d (not a multiple of 16 / 128), ragged chain tiles (C = 130: two tiles of 128), one and several leapfrog steps"""
    prec = synth.dense_gaussian_precision(d, seed=d % 89)
    init = synth.initial_states(C, d, seed=d + 1) * 0.5
    st = _settings(algo, 3, 2, 4, L, EPS[algo])
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("gemm_step_kernel<"), kern
    o_draws, o = _oracle(algo, d, prec, init, 3, 2, 4, L, EPS[algo], chain0=11)
    assert 0 < o["n_accept"].sum()
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.array_equal(g["theta"], o_draws[-1])
    if algo == "hmc":
        assert np.array_equal(g["n_leap"], o["n_leap"])


@pytest.mark.parametrize("algo", ["hmc", "mala"])
def test_matrix_product_samplers_in_the_non_finite_regime(algo):
    """step sizes that blow chains up and initial values that are +-inf / NaN already: flagged by the accept step, replayed literally"""
    d, C = 600, 40
    prec = synth.dense_gaussian_precision(d, seed=7)
    init = synth.initial_states(C, d, seed=d) * 0.5
    init[3] *= 1e200; init[7, 5] = np.inf; init[20, d - 1] = np.nan; init[33] *= 1e160
    for eps in (EPS[algo], 1e6):
        st = _settings(algo, 5, 2, 3, 3, eps)
        g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
        assert mcmc_amd.last_kernel().startswith("gemm_step_kernel<")
        o_draws, o = _oracle(algo, d, prec, init, 5, 2, 3, 3, eps)
        assert np.array_equal(g["n_accept"], o["n_accept"]), eps
        assert np.array_equal(g_draws, o_draws, equal_nan=True), eps
        assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True), eps


@pytest.mark.parametrize("algo", ["hmc", "mala", "rwmh"])
def test_matrix_product_samplers_equal_the_literal_kernel_on_more_chains(algo):
    """three chain tiles (one ragged) x six row tiles, 12 draws: every (row tile, chain tile) pair and the accepted-state bookkeeping over many draws;
    the literal kernel (one workgroup per chain, the reference's operations as written) is what ran these shapes before"""
    d, C = 700, 300
    prec = synth.dense_gaussian_precision(d, seed=3)
    init = synth.initial_states(C, d, seed=9) * 0.4
    st = _settings(algo, 21, 4, 8, 5, EPS[algo])
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=1000)
    assert mcmc_amd.last_kernel().startswith("gemm_step_kernel<")
    l_draws, l = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=1000, kernel_hint=mcmc_amd.KERNEL_LITERAL)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<")
    assert 0 < l["n_accept"].sum() < 8 * C
    assert np.array_equal(g["n_accept"], l["n_accept"]) and np.array_equal(g_draws, l_draws) and np.array_equal(g["theta"], l["theta"])


def test_matrix_product_hmc_continues_a_run():
    """a run cut into two calls (mi_chains.draw0) equals the run in one piece: the Philox counters take the global draw index"""
    d, C = 520, 33
    prec = synth.dense_gaussian_precision(d, seed=11)
    init = synth.initial_states(C, d, seed=4) * 0.5
    whole, w = mcmc_amd.sample("hmc", mcmc_amd.TARGET_GAUSS_DENSE, init, _settings("hmc", 8, 0, 6, 3, 0.02), prec=prec)
    a, ga = mcmc_amd.sample("hmc", mcmc_amd.TARGET_GAUSS_DENSE, init, _settings("hmc", 8, 0, 2, 3, 0.02), prec=prec)
    b, gb = mcmc_amd.sample("hmc", mcmc_amd.TARGET_GAUSS_DENSE, np.ascontiguousarray(ga["theta"].T), _settings("hmc", 8, 0, 4, 3, 0.02), prec=prec, draw0=2)
    assert np.array_equal(whole, np.concatenate([a, b]))
    assert np.array_equal(w["n_accept"], ga["n_accept"] + gb["n_accept"])


def test_matrix_product_hmc_recovers_the_covariance():
    """statistical check at d = 576: per-dimension variances of 4096 hmc chains against diag(P^-1) (the launches of a draw replayed from a graph)"""
    d, C = 576, 4096
    prec = synth.dense_gaussian_precision(d, seed=5)
    cov = np.linalg.inv(prec)
    rng = np.random.default_rng(1)
    init = rng.multivariate_normal(np.zeros(d), cov, size=C)
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=30, n_keep_draws=1, n_leap_steps=8, step_size=0.12)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert mcmc_amd.last_kernel().startswith("gemm_step_kernel<")
    v = g_draws[0].var(axis=1)
    assert np.all(np.abs(v / np.diag(cov) - 1.0) < 0.15)
    assert g["n_accept"].mean() > 0.5


# ---- the logistic-regression target beyond d = 512: eta = X Theta and X^T (y - sigmoid(eta)) as two matrix products per gradient, the row terms in between
# element-wise (gemm_rowterm_kernel); the oracle's plain orders (one eta chain per row, rows ascending in the gradient, four-strided sums)
def _logit_oracle(algo, d, X, y, init, seed, burn, keep, L, eps, chain0=0):
    s = orc.make_settings(seed=seed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1)
    return orc.run_many(ALGO[algo], orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4), init, s, chain0=chain0)


LEPS = {"hmc": 0.02, "mala": 0.03, "rwmh": 0.01}


@pytest.mark.parametrize("algo", ["hmc", "mala", "rwmh"])
@pytest.mark.parametrize("d,N,C,L", [(513, 40, 45, 3), (640, 300, 130, 1), (1030, 129, 20, 2)])
def test_logistic_beyond_d512_equals_the_oracle(algo, d, N, C, L):
    X, y = synth.logistic_problem(d, N, seed=5)
    init = synth.initial_states(C, d, seed=8) * 0.1
    st = _settings(algo, 12, 2, 4, L, LEPS[algo])
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=5)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("gemm_step_kernel<") and ", 1>" in kern, kern
    o_draws, o = _logit_oracle(algo, d, X, y, init, 12, 2, 4, L, LEPS[algo], chain0=5)
    assert 0 < o["n_accept"].sum()
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.array_equal(g["theta"], o_draws[-1])


@pytest.mark.parametrize("algo", ["hmc", "mala"])
def test_logistic_beyond_d512_in_the_non_finite_regime(algo):
    d, N, C = 600, 64, 40
    X, y = synth.logistic_problem(d, N, seed=3)
    init = synth.initial_states(C, d, seed=d) * 0.1
    init[3] *= 1e200; init[7, 5] = np.inf; init[20, d - 1] = np.nan; init[33] *= 1e160
    for eps in (LEPS[algo], 1e6):
        st = _settings(algo, 5, 2, 3, 3, eps)
        g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
        assert mcmc_amd.last_kernel().startswith("gemm_step_kernel<")
        o_draws, o = _logit_oracle(algo, d, X, y, init, 5, 2, 3, 3, eps)
        assert np.array_equal(g["n_accept"], o["n_accept"]), eps
        assert np.array_equal(g_draws, o_draws, equal_nan=True), eps
        assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True), eps


@pytest.mark.parametrize("algo", ["hmc", "mala", "rwmh"])
def test_logistic_beyond_d512_equals_the_literal_kernel_on_more_chains(algo):
    d, N, C = 700, 200, 300
    X, y = synth.logistic_problem(d, N, seed=9)
    init = synth.initial_states(C, d, seed=9) * 0.1
    st = _settings(algo, 21, 4, 8, 4, LEPS[algo])
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=1000)
    assert mcmc_amd.last_kernel().startswith("gemm_step_kernel<")
    l_draws, l = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=1000, kernel_hint=mcmc_amd.KERNEL_LITERAL)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<")
    assert 0 < l["n_accept"].sum() <= 8 * C
    assert np.array_equal(g["n_accept"], l["n_accept"]) and np.array_equal(g_draws, l_draws) and np.array_equal(g["theta"], l["theta"])


# ---- a DIAGONAL precond_mat beyond d = 512 (ref: src/hmc.cpp:57-59,158-160,171,184: p = sqrt(m) z, theta += eps (p / m), K = p.(p / m) / 2; src/mala.cpp:57-58,123,159 with
# include/mcmc/mala.ipp:58-64: mean = x + eps^2 (m grad) / 2, noise eps sqrt(m) z, INV(eps^2 M) diagonal): element-wise in the same kernels (tables of ones for the identity)
@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("target,d", [("dense", 600), ("dense", 1024), ("logit", 513), ("logit", 700)])
def test_matrix_product_samplers_with_a_diagonal_precond_mat(algo, target, d):
    C = 40
    M = np.diag(np.random.default_rng(d + 1).uniform(0.4, 2.5, d))
    eps = 0.02 if algo == "hmc" else 0.03
    init = synth.initial_states(C, d, seed=d + 2) * (0.5 if target == "dense" else 0.1)
    init[5] *= 1e200; init[9, 3] = np.inf                 # two chains leave the finite regime: replayed literally with the same tables
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=2, n_keep_draws=5, n_leap_steps=3, step_size=eps, precond_mat=M)
    s = orc.make_settings(seed=7, n_burnin=2, n_keep=5, n_leap=3, step=eps, W=4, hoist=1, precond=M)
    if target == "dense":
        prec = synth.dense_gaussian_precision(d, seed=d % 89)
        g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    else:
        X, y = synth.logistic_problem(d, 70, seed=5)
        g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
        t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("gemm_step_kernel<") and "diagonal precond_mat" in kern, kern
    o_draws, o = orc.run_many(ALGO[algo], t, init, s)
    assert o["n_accept"].sum() > 0
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


def test_rwmh_with_a_cov_mat_beyond_d512_stays_on_the_literal_kernel():
    d, C = 520, 6
    prec = synth.dense_gaussian_precision(d, seed=3)
    M = np.diag(np.random.default_rng(1).uniform(0.5, 2.0, d))
    init = synth.initial_states(C, d, seed=2) * 0.5
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=1, n_keep_draws=3, step_size=0.01, precond_mat=M)
    mcmc_amd.sample("rwmh", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<3>"), mcmc_amd.last_kernel()
