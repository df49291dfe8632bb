"""GPU: BASELINE.json configs 3, 4, 5 at their full sizes -- the oracle re-runs a sample of the chains (by
global chain id) bit-exactly; the population is checked through properties the domain offers."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def test_config3_mala_logistic_d512_262144_chains():
    d, N, C = 512, 1024, 262144
    X, y = synth.logistic_problem(d, N)
    init = np.zeros((C, d))
    init[:, 0] = np.linspace(-0.5, 0.5, C)                 # distinct starts
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=3, n_keep_draws=2, step_size=0.02)
    draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    assert draws.shape == (2, d, C) and np.isfinite(draws).all()
    pick = [0, 17, 4095, 131072, 262143]
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=128, eta_chains=2)
    for c in pick:
        s = orc.make_settings(seed=6, n_burnin=3, n_keep=2, step=0.02, W=4, hoist=1, blocks=4, block_size=128, chain_id=c)
        o, info = orc.run_chain(orc.ALGO_MALA, t, init[c], s)
        assert np.array_equal(draws[:, :, c], o) and g["n_accept"][c] == info["n_accept"]
    assert g["n_accept"].mean() / 2 > 0.9                   # eps = 0.02 at d = 512: nearly every move accepted


def test_config4_nuts_d128_65536_chains():
    d, C = 128, 65536
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=3)
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=6, n_keep_draws=3, n_adapt_draws=6, max_tree_depth=10)
    draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert draws.shape == (3, d, C) and np.isfinite(draws).all()
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    for c in [0, 15, 16, 40000, 65535]:
        s = orc.make_settings(seed=4, n_burnin=6, n_keep=3, n_adapt=6, step=1.0, W=4, chain_id=c)
        o, info = orc.run_chain(orc.ALGO_NUTS, t, init[c], s, traces=True)
        assert np.array_equal(draws[:, :, c], o)
        assert np.array_equal(g["depth"][:, c], info["depth"]) and g["n_leap"][c] == info["n_leap"]
        assert g["eps"][c] == info["eps"]
    assert g["depth"].max() <= 10 and g["n_leap"].min() >= 9


def test_config5_hmc_d1024_diag_one_gpu_shard():
    d, C = 1024, 131072                                     # one GPU's share of 2^20 chains
    prec = synth.ill_conditioned_diag(d, 1.0e4)
    chain0 = 3 * C                                          # as rank 3 of 8 would run it
    init = (synth.initial_states(C, d, seed=3, chain0=chain0) / np.sqrt(prec)[None, :])
    st = mcmc_amd.default_settings(rng_seed_value=8, n_burnin_draws=2, n_keep_draws=2, n_leap_steps=32, step_size=0.005)
    draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec, chain0=chain0)
    assert draws.shape == (2, d, C) and np.isfinite(draws).all()
    t = orc.TargetSpec(orc.TARGET_DIAG, d, prec=prec, W=4)
    for c in [0, 1, 77777, C - 1]:
        s = orc.make_settings(seed=8, n_burnin=2, n_keep=2, n_leap=32, step=0.005, W=4, chain_id=chain0 + c)
        o, info = orc.run_chain(orc.ALGO_HMC, t, init[c], s)
        assert np.array_equal(draws[:, :, c], o) and g["n_accept"][c] == info["n_accept"]
    # started in stationarity: the scaled second moment stays 1
    m2 = (draws[-1] ** 2 * prec[:, None]).mean()
    assert abs(m2 - 1) < 0.02
