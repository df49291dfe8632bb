"""GPU: mi_mcmc_hmc_run_mass_adapted -- hmc with a diagonal mass matrix pooled over the chains (NOT a reference mode).
What is checked: the run is a chain of ordinary hmc calls (the last part reproduces bit for bit from the reported mass and the
oracle), it is reproducible, and it does what it is for: the ill-conditioned target of BASELINE configs[4] mixes."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth
from mcmc_amd.ess import ess_per_chain

pytestmark = pytest.mark.gpu


def _run(kind, d, C, prec, init, n_windows, burn, keep, L, eps, seed=5):
    t = mcmc_amd.make_target(kind, d, prec=prec)
    theta = np.ascontiguousarray(init.T.copy())
    draws = np.zeros((keep, d, C))
    nacc = np.zeros(C, dtype=np.uint64)
    st = mcmc_amd.default_settings(rng_seed_value=seed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
    ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc)
    mass = mcmc_amd.hmc_mass_adapted(t, st, ch, n_windows=n_windows)
    return draws, nacc, mass, theta


def test_a_run_is_ordinary_hmc_with_the_reported_mass_bit_exact_vs_oracle():
    """No re-estimation window (n_windows = 0): the mass comes from the spread of initial_vals and the whole run is ONE ordinary
    mcmc::hmc with precond_mat = diag(mass) -- reproduced bit for bit by the oracle from the reported mass."""
    d, C, burn, keep, L, eps = 24, 200, 9, 12, 6, 0.25
    prec = synth.ill_conditioned_diag(d, 400.0)
    init = synth.initial_states(C, d, seed=2) / np.sqrt(prec)[None, :] * 1.7
    draws, nacc, mass, _ = _run(mcmc_amd.TARGET_GAUSS_DIAG, d, C, prec, init, 0, burn, keep, L, eps)
    assert np.all(mass > 0) and np.isfinite(mass).all()
    assert np.allclose(mass, 1.0 / init.var(axis=0, ddof=1), rtol=1e-12)     # pooled over the chains, per dimension
    t = orc.TargetSpec(orc.TARGET_DIAG, d, prec=prec, W=4)
    s = orc.make_settings(seed=5, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, precond=np.diag(mass))
    o, info = orc.run_many(orc.ALGO_HMC, t, init, s)
    assert np.array_equal(draws, o) and np.array_equal(nacc, info["n_accept"])
    # with re-estimation windows the parts chain through draw0: every later estimate sees the states the earlier parts left
    d2, n2, m2, _ = _run(mcmc_amd.TARGET_GAUSS_DIAG, d, C, prec, init, 2, 30, keep, L, eps)
    assert np.allclose(m2, prec, rtol=0.5) and not np.array_equal(m2, mass) and n2.sum() > 0


def test_mass_adaptation_makes_the_ill_conditioned_target_mix():
    """BASELINE configs[4] target (d = 1024, precisions 1 .. 1e4).  With precond_mat = I the step size the stiffest dimension
    allows (0.005) freezes the soft ones; the pooled diagonal mass runs every dimension at unit frequency with step 0.12."""
    d, C, burn, keep, L = 1024, 2048, 40, 60, 32
    prec = synth.ill_conditioned_diag(d, 1.0e4)
    init = synth.initial_states(C, d, seed=3) / np.sqrt(prec)[None, :]
    draws, nacc, mass, _ = _run(mcmc_amd.TARGET_GAUSS_DIAG, d, C, prec, init, 3, burn, keep, L, 0.12)
    assert np.allclose(mass, prec, rtol=0.25)
    assert nacc.mean() / keep > 0.8
    ess_adapted = ess_per_chain(draws[:, ::37, :]).min()
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.005)
    plain, _ = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec)
    ess_plain = ess_per_chain(plain[:, ::37, :]).min()
    print(f"min ESS per chain over {keep} draws: identity mass {ess_plain:.2f}, pooled diagonal mass {ess_adapted:.1f}")
    assert ess_adapted > 20 * ess_plain and ess_adapted > 0.3 * keep
    m2 = (draws[-1] ** 2 * prec[:, None]).mean()
    assert abs(m2 - 1) < 0.05                                       # still the right distribution


def test_mass_adapted_run_is_reproducible_and_rejects_a_user_precond():
    d, C = 16, 64
    prec = synth.dense_gaussian_precision(d, seed=3)
    init = synth.initial_states(C, d, seed=1)
    a = _run(mcmc_amd.TARGET_GAUSS_DENSE, d, C, prec, init, 2, 12, 6, 4, 0.3)
    b = _run(mcmc_amd.TARGET_GAUSS_DENSE, d, C, prec, init, 2, 12, 6, 4, 0.3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec)
    st = mcmc_amd.default_settings(n_burnin_draws=2, n_keep_draws=2, precond_mat=np.eye(d))
    with pytest.raises(mcmc_amd.MiMcmcError):
        mcmc_amd.hmc_mass_adapted(t, st, mcmc_amd.make_chains(np.ascontiguousarray(init.T.copy()), C))
