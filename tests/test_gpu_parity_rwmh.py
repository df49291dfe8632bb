"""GPU: mcmc::rwmh (many chains) vs the CPU oracle through the C ABI -- bit-exact.  SURVEY 8 (f-4).
settings.step_size carries rwmh_settings.par_scale, settings.precond_mat carries rwmh_settings.cov_mat."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _bounds(d, seed=0):
    rng = np.random.default_rng(seed)
    kind = rng.integers(1, 5, d)
    return np.where((kind == 2) | (kind == 4), -1.5, -np.inf), np.where((kind == 3) | (kind == 4), 2.0, np.inf)


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.5, 1.5, d))


CASES = [
    # kind,  d,   C, scale, burn, keep, cov,     bounded
    ("dense", 8, 16, 0.30, 5, 20, None, False),
    ("iso", 3, 7, 0.80, 10, 30, None, False),
    ("dense", 128, 64, 0.05, 4, 10, None, False),
    ("dense", 100, 130, 0.10, 2, 6, "diag", False),
    ("diag", 40, 48, 0.20, 3, 9, "diag", True),
    ("dense", 64, 33, 0.15, 2, 8, "dense", False),
    ("dense", 128, 40, 0.08, 2, 8, "dense", False),  # d > 64: par_scale * chol(cov_mat) read from L2 in fragment order
    ("dense", 100, 20, 0.08, 1, 6, "dense", True),
    ("dense", 20, 24, 0.30, 3, 12, "dense", True),
    ("dense", 37, 20, 0.10, 0, 10, None, True),      # ragged d, bounds, no burn-in
]


@pytest.mark.parametrize("kind,d,C,scale,burn,keep,cov,bounded", CASES)
def test_rwmh_bit_exact_vs_oracle(kind, d, C, scale, burn, keep, cov, bounded):
    init = np.clip(synth.initial_states(C, d, seed=31) * 0.4, -1.0, 1.5)
    prec, k_gpu, k_orc = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    if kind == "dense":
        prec, k_gpu, k_orc = synth.dense_gaussian_precision(d, seed=7), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
    elif kind == "diag":
        prec, k_gpu, k_orc = synth.ill_conditioned_diag(d, 30.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    kw, okw = {}, {}
    if cov == "diag":
        M = np.diag(np.linspace(0.5, 2.0, d)); kw.update(precond_mat=M); okw.update(precond=M)
    if cov == "dense":
        M = _spd(d, seed=d); kw.update(precond_mat=M); okw.update(precond=M)
    if bounded:
        lb, ub = _bounds(d, seed=d)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=burn, n_keep_draws=keep, step_size=scale, **kw)
    g_draws, g = mcmc_amd.rwmh(k_gpu, init, st, prec=prec, chain0=77)
    t = orc.TargetSpec(k_orc, d, prec=prec, W=4)
    s = orc.make_settings(seed=99, n_burnin=burn, n_keep=keep, step=scale, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RWMH, t, init, s, chain0=77)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    if bounded:
        assert ((g_draws >= lb[None, :, None]) & (g_draws <= ub[None, :, None])).all()
    assert 0 < g["n_accept"].sum() <= C * keep


def test_rwmh_resumes_bit_exactly():
    d, C = 24, 20
    prec = synth.dense_gaussian_precision(d, seed=2)
    init = synth.initial_states(C, d, seed=8) * 0.5
    mk = lambda b, k: mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=b, n_keep_draws=k, step_size=0.2)
    w_draws, w = mcmc_amd.rwmh(mcmc_amd.TARGET_GAUSS_DENSE, init, mk(2, 9), prec=prec)
    a_draws, a = mcmc_amd.rwmh(mcmc_amd.TARGET_GAUSS_DENSE, init, mk(2, 4), prec=prec)
    b_draws, b = mcmc_amd.rwmh(mcmc_amd.TARGET_GAUSS_DENSE, a["theta"].T.copy(), mk(0, 5), prec=prec, draw0=6)
    assert np.array_equal(np.concatenate([a_draws, b_draws]), w_draws)


# ---------------------------------------------------------------- logistic-regression target (logit_lds_kernel<., RWMH>)
LOGIT_CASES = [
    # d,   N,    C,  par_scale, burn, keep
    (5, 40, 16, 0.30, 5, 20),        # SURVEY 8(c) golden shape "logistic d=5"
    (64, 100, 20, 0.05, 3, 8),       # one tile per wave, ragged N and C
    (100, 333, 40, 0.03, 2, 6),      # d_pad 128, ragged everything
    (512, 1024, 32, 0.01, 2, 4),     # config 3 dimensions
    (300, 64, 17, 0.05, 0, 10),
]


@pytest.mark.parametrize("d,N,C,scale,burn,keep", LOGIT_CASES)
def test_rwmh_logistic_bit_exact_vs_oracle(d, N, C, scale, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=scale)
    g_draws, g = mcmc_amd.rwmh(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    dq = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=dq, eta_chains=2)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=scale, W=4, blocks=4, block_size=dq)
    o_draws, o = orc.run_many(orc.ALGO_RWMH, t, init, s, chain0=3)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert 0 < int(g["n_accept"].sum()) < C * keep
