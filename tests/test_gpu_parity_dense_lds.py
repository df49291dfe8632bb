"""GPU: dense Gaussian targets with 128 < d <= 512 (ref: src/hmc.cpp:155-205, src/mala.cpp:149-186, src/rwmh.cpp:123-151 --
n_vals is unrestricted).  P does not fit into LDS beyond d = 128, so hmc / mala / rwmh stream it through LDS block by block on the matrix
cores (mcmc_amd/csrc/logistic_lds.hpp, LOGIT_TARGET_DENSE).  Bit for bit against the oracle: rows of P x as one ascending fma chain,
dot products over the kernel's four dimension quarters; chains that reach the non-finite regime are replayed literally."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu
ALGO = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "rwmh": orc.ALGO_RWMH}


def _blk(d):
    return dict(blocks=4, block_size=48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128)     # 16 dims x tiles per wave


def _run(algo, d, C, eps, init, seed=3, burn=2, keep=5, L=4, chain0=0, draw0=0):
    prec = synth.dense_gaussian_precision(d, seed=d % 89)
    st = mcmc_amd.default_settings(rng_seed_value=seed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=chain0)
    kern = mcmc_amd.last_kernel()
    s = orc.make_settings(seed=seed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1, **_blk(d))
    o_draws, o = orc.run_many(ALGO[algo], orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **_blk(d)), init, s, chain0=chain0)
    return g_draws, g, o_draws, o, kern


@pytest.mark.parametrize("algo,eps", [("hmc", 0.04), ("mala", 0.06), ("rwmh", 0.03)])
@pytest.mark.parametrize("d", [129, 192, 193, 256, 257, 300, 384, 400, 512])
def test_dense_gaussian_streamed_through_lds_equals_the_oracle(algo, eps, d):
    C = 45                                              # one full workgroup of 32 chains + a ragged one
    init = synth.initial_states(C, d, seed=d + 1) * 0.5
    g_draws, g, o_draws, o, kern = _run(algo, d, C, eps, init, chain0=11)
    assert kern.startswith("logit_lds_kernel<") and kern.split(",")[2].strip() == "1", kern        # TARGET = dense Gaussian
    assert 0 < o["n_accept"].sum()
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.array_equal(g["theta"], o_draws[-1])
    if algo == "hmc":
        assert np.array_equal(g["n_leap"], o["n_leap"])


@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("d", [160, 512])
def test_dense_gaussian_beyond_d128_in_the_non_finite_regime(algo, d):
    """step sizes that blow chains up and initial values that are +-inf / NaN already: flagged by the streamed kernel, replayed literally"""
    C = 40
    init = synth.initial_states(C, d, seed=d) * 0.5
    init[3] *= 1e200; init[7, 5] = np.inf; init[20, d - 1] = np.nan; init[33] *= 1e160
    for eps in (0.05, 1e6):
        g_draws, g, o_draws, o, kern = _run(algo, d, C, eps, init, keep=3, L=3)
        assert np.array_equal(g["n_accept"], o["n_accept"]), eps
        assert np.array_equal(g_draws, o_draws, equal_nan=True), eps
        assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True), eps


def test_dense_gaussian_beyond_d128_recovers_the_covariance():
    """statistical check at d = 192: per-dimension variances of 4096 hmc chains against diag(P^-1)"""
    d, C = 192, 4096
    prec = synth.dense_gaussian_precision(d, seed=5)
    cov = np.linalg.inv(prec)
    rng = np.random.default_rng(1)
    init = rng.multivariate_normal(np.zeros(d), cov, size=C)
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=30, n_keep_draws=1, n_leap_steps=8, step_size=0.15)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    v = g_draws[0].var(axis=1)                         # [d] over chains
    assert np.all(np.abs(v / np.diag(cov) - 1.0) < 0.15)
    assert g["n_accept"].mean() > 0.5


# ---- hmc with a DIAGONAL precond_mat alone on the LDS-streamed kernels (their DIAGM instantiation; ref: src/hmc.cpp:57-59,158-160,171,184):
# the dense Gaussian of this file and the logistic-regression target they were written for
@pytest.mark.parametrize("d", [160, 256, 400, 512])
def test_dense_gaussian_hmc_with_a_diagonal_precond_mat(d):
    C = 40
    prec = synth.dense_gaussian_precision(d, seed=d % 89)
    M = np.diag(np.random.default_rng(d).uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=d + 1) * 0.5
    init[5] *= 1e200; init[9, 3] = np.inf                 # two chains leave the finite regime: replayed literally with the same matrix
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=2, n_keep_draws=5, n_leap_steps=4, step_size=0.04, precond_mat=M)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.endswith("1, true>"), kern
    s = orc.make_settings(seed=3, n_burnin=2, n_keep=5, n_leap=4, step=0.04, W=4, hoist=1, precond=M, **_blk(d))
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **_blk(d)), init, s, chain0=11)
    assert o["n_accept"].sum() > 0
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


@pytest.mark.parametrize("d,N", [(9, 20), (40, 33), (130, 64), (512, 100)])
def test_logistic_hmc_with_a_diagonal_precond_mat(d, N):
    C = 40
    X, y = synth.logistic_problem(d, N, seed=5)
    M = np.diag(np.random.default_rng(d).uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=8) * 0.3
    init[5] *= 1e200; init[9, 3] = np.inf
    bs = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    st = mcmc_amd.default_settings(rng_seed_value=12, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=3, step_size=0.1, precond_mat=M)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.endswith("0, true>"), kern
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=bs, eta_chains=2)
    s = orc.make_settings(seed=12, n_burnin=2, n_keep=4, n_leap=3, step=0.1, W=4, hoist=1, precond=M, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s)
    assert o["n_accept"].sum() > 0
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


@pytest.mark.parametrize("target,d", [("dense", 160), ("dense", 512), ("logit", 9), ("logit", 130), ("logit", 512)])
def test_mala_with_a_diagonal_precond_mat_on_the_streamed_kernels(target, d):
    """ref: src/mala.cpp:57-58,123,159 with include/mcmc/mala.ipp:58-64: mean = x + eps^2 (M grad) / 2, noise eps sqrt(M) z, dmvnorm with
    Sigma = eps^2 M (INV and LOG_DET from the host in the oracle's operation order) -- M diagonal, no bounds."""
    C = 40
    M = np.diag(np.random.default_rng(d + 1).uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=d + 2) * 0.3
    init[5] *= 1e200; init[9, 3] = np.inf
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=2, n_keep_draws=6, step_size=0.05, precond_mat=M)
    if target == "dense":
        prec = synth.dense_gaussian_precision(d, seed=d % 89)
        g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **_blk(d)); blk = _blk(d)
    else:
        X, y = synth.logistic_problem(d, 40, seed=5)
        g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
        bs = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
        blk = dict(blocks=4, block_size=bs)
        t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, eta_chains=2, **blk)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.endswith("true>"), kern
    s = orc.make_settings(seed=7, n_burnin=2, n_keep=6, step=0.05, W=4, hoist=1, precond=M, **blk)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s)
    assert o["n_accept"].sum() > 0
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


# ---- hmc with a DENSE precond_mat on the LDS-streamed kernels (round 5; their DENSEM instantiation; ref: src/hmc.cpp:57-59 -- inv_precond_matrix =
# INV(M), sqrt_precond_matrix = CHOL_LOWER(M) --, :158 L z, :160,184 p.(Minv p) / 2, :171 theta += eps (Minv p)): both matrices streamed through
# LDS like P, every product the oracle's ascending fma chain; no bounds (with them: literal.hpp)
def _dense_m(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))


@pytest.mark.parametrize("d", [129, 160, 192, 200, 256, 300, 384, 512])
def test_dense_gaussian_hmc_with_a_dense_precond_mat(d):
    C = 45                                              # one full workgroup of 32 chains + a ragged one
    prec = synth.dense_gaussian_precision(d, seed=d % 89)
    M = _dense_m(d, d)
    init = synth.initial_states(C, d, seed=d + 1) * 0.5
    init[5] *= 1e200; init[9, 3] = np.inf                 # two chains leave the finite regime: replayed literally with the same matrices
    for L in (3, 0):                                    # (n_leap_steps = 0: the two kinetic products back to back)
        st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=L, step_size=0.3, precond_mat=M)
        g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11)
        kern = mcmc_amd.last_kernel()
        assert kern.startswith("logit_lds_kernel<") and kern.endswith("1, false, false, true>"), kern
        s = orc.make_settings(seed=3, n_burnin=2, n_keep=4, n_leap=L, step=0.3, W=4, hoist=1, precond=M, **_blk(d))
        o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **_blk(d)), init, s, chain0=11)
        assert L == 0 or 0 < o["n_accept"].sum() < 4 * C
        assert np.array_equal(g["n_accept"], o["n_accept"]), L
        assert np.array_equal(g_draws, o_draws, equal_nan=True), L
        assert np.array_equal(g["n_leap"], o["n_leap"]), L


@pytest.mark.parametrize("d,N", [(9, 20), (40, 33), (100, 64), (130, 64), (300, 50), (512, 100)])
def test_logistic_hmc_with_a_dense_precond_mat(d, N):
    C = 40
    X, y = synth.logistic_problem(d, N, seed=5)
    M = _dense_m(d, d + 7)
    init = synth.initial_states(C, d, seed=8) * 0.3
    init[5] *= 1e200; init[9, 3] = np.inf
    bs = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    st = mcmc_amd.default_settings(rng_seed_value=12, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=3, step_size=0.1, precond_mat=M)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.endswith("0, false, false, true>"), kern
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=bs, eta_chains=2)
    s = orc.make_settings(seed=12, n_burnin=2, n_keep=4, n_leap=3, step=0.1, W=4, hoist=1, precond=M, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s)
    assert 0 < o["n_accept"].sum() <= 4 * C
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


def test_dense_precond_mat_with_bounds_stays_on_the_literal_kernel():
    d, C = 160, 6
    prec = synth.dense_gaussian_precision(d, seed=3)
    M = _dense_m(d, 1)
    lb = np.full(d, -np.inf); ub = np.full(d, np.inf); lb[::3] = -1.5; ub[::5] = 2.0
    init = np.clip(synth.initial_states(C, d, seed=4) * 0.5, -1.0, 1.5)
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=1, n_keep_draws=3, n_leap_steps=3, step_size=0.04, precond_mat=M,
                                   vals_bound=1, lower_bounds=lb, upper_bounds=ub)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<0>"), mcmc_amd.last_kernel()
    s = orc.make_settings(seed=5, n_burnin=1, n_keep=3, n_leap=3, step=0.04, W=4, hoist=1, precond=M, lower=lb, upper=ub, **_blk(d))
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **_blk(d)), init, s)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)


# ---- mala with a DENSE precond_mat on the same kernels (ref: src/mala.cpp:57-58,123,159; include/mcmc/mala.ipp:58-64; include/stats/dmvnorm.hpp:37-41):
# M grad, L z and the two INV(eps^2 M) (X - mu) streamed through LDS; INV / LOG_DET of the constant Sigma from the host
@pytest.mark.parametrize("target,d", [("dense", 129), ("dense", 192), ("dense", 256), ("dense", 300), ("dense", 512),
                                      ("logit", 9), ("logit", 40), ("logit", 100), ("logit", 200), ("logit", 512)])
def test_mala_with_a_dense_precond_mat_on_the_streamed_kernels(target, d):
    C = 45
    M = _dense_m(d, d + 3)
    init = synth.initial_states(C, d, seed=d + 2) * 0.3
    init[5] *= 1e200; init[9, 3] = np.inf                 # flagged, replayed literally with the same matrices
    eps = 0.12 if target == "dense" else 0.25
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=2, n_keep_draws=5, step_size=eps, precond_mat=M)
    if target == "dense":
        prec = synth.dense_gaussian_precision(d, seed=d % 89)
        g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=5)
        blk = _blk(d); t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, **blk)
    else:
        X, y = synth.logistic_problem(d, 40, seed=5)
        g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=5)
        blk = dict(blocks=4, block_size=16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128)
        t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, eta_chains=2, **blk)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.split(",")[1].strip() == "0" and kern.endswith("false, false, true>"), kern
    s = orc.make_settings(seed=7, n_burnin=2, n_keep=5, step=eps, W=4, hoist=1, precond=M, **blk)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=5)
    assert 0 < o["n_accept"].sum() < 5 * C
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws, equal_nan=True)
