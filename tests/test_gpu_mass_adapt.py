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


# ---- per-chain diagonal masses (mi_chains.mass_diag) and their adaptation (mi_mcmc_hmc_run_mass_adapted_per_chain): SURVEY 8 f-2's own
# wording; ref: src/hmc.cpp:57-59,158-160,171,184 is what each chain does with ITS precond_mat

def _oracle_per_chain(kind, d, prec, init, mass, seed, burn, keep, L, eps, **okw):
    """chain c = mcmc::hmc with precond_mat = diag(mass[:, c])"""
    C = init.shape[0]
    draws = np.zeros((keep, d, C)); nacc = np.zeros(C, dtype=np.uint64)
    t = orc.TargetSpec(kind, d, prec=prec, W=4)
    for c in range(C):
        s = orc.make_settings(seed=seed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, precond=np.diag(mass[:, c]), **okw)
        o, info = orc.run_many(orc.ALGO_HMC, t, init[c:c + 1], s, chain0=c)
        draws[:, :, c] = o[:, :, 0]; nacc[c] = info["n_accept"][0]
    return draws, nacc


@pytest.mark.parametrize("case", ["diag_elementwise", "iso_elementwise_d1", "dense_mfma_d20", "dense_mfma_d100",
                                  pytest.param("dense_mfma_d100_wide", marks=pytest.mark.gpu_slow), "dense_bounded_literal", "diag_bounded_literal"])
def test_per_chain_masses_are_c_calls_of_hmc_with_their_own_precond_mat(case):
    rng = np.random.default_rng(7)
    burn, keep, L, eps, C = 3, 6, 5, 0.2, 70
    kw, okw = {}, {}
    if case == "diag_elementwise":
        d = 37; prec = synth.ill_conditioned_diag(d, 100.0); kg, ko = mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    elif case == "iso_elementwise_d1":
        d = 1; prec = None; kg, ko = mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    elif case.startswith("dense_mfma"):                      # round 4: per-chain tables on the MFMA kernel (hmc_gauss_mfma_kernel<., ., false, false, true, true>)
        d = 20 if case.endswith("d20") else 100; C = 9 if d == 20 else (150 if case.endswith("wide") else 40)     # (the oracle runs chain by chain: 150 chains are 18 s)
        prec = synth.dense_gaussian_precision(d, seed=4); kg, ko = mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
    elif case == "dense_bounded_literal":
        d = 20; C = 9; prec = synth.dense_gaussian_precision(d, seed=4); kg, ko = mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
        lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 0, 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    else:
        d = 12; C = 9; prec = synth.ill_conditioned_diag(d, 30.0); kg, ko = mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
        lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 0, 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    init = np.clip(synth.initial_states(C, d, seed=3) * 0.5, -1.0, 1.5)
    mass = np.ascontiguousarray(rng.uniform(0.2, 5.0, (d, C)))
    if case in ("diag_elementwise", "dense_mfma_d100", "dense_mfma_d100_wide"): init[1, 0] = 1e200      # one chain of the throughput kernels leaves the finite regime: replayed literally
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps, **kw)
    t = mcmc_amd.make_target(kg, d, prec=prec)
    theta = np.ascontiguousarray(init.T.copy()); draws = np.zeros((keep, d, C)); nacc = np.zeros(C, dtype=np.uint64)
    mcmc_amd.run("hmc", t, st, mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mass_diag=mass))
    want = "hmc_diag" if "elementwise" in case else "hmc_gauss_mfma_kernel<" if "mfma" in case else "literal_kernel<0>"
    assert mcmc_amd.last_kernel().startswith(want), mcmc_amd.last_kernel()
    if "mfma" in case: assert mcmc_amd.last_kernel().endswith("true, true>")
    o_draws, o_nacc = _oracle_per_chain(ko, d, prec, init, mass, 9, burn, keep, L, eps, **okw)
    assert o_nacc.sum() > 0
    assert np.array_equal(nacc, o_nacc)
    assert np.array_equal(draws, o_draws, equal_nan=True)
    # the other samplers refuse the field instead of ignoring it; so does hmc next to a precond_mat
    with pytest.raises(mcmc_amd.MiMcmcError):
        mcmc_amd.run("mala", t, st, mcmc_amd.make_chains(theta.copy(), C, mass_diag=mass))
    if not kw:
        st2 = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1, precond_mat=np.eye(d))
        with pytest.raises(mcmc_amd.MiMcmcError):
            mcmc_amd.run("hmc", t, st2, mcmc_amd.make_chains(theta.copy(), C, mass_diag=mass))


def _adapt_per_chain(t, init, d, C, burn, keep, L, eps0, n_windows, first=0.0):
    theta = np.ascontiguousarray(init.T.copy()); draws = np.zeros((keep, d, C)); nacc = np.zeros(C, dtype=np.uint64)
    mass = np.zeros((d, C))
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps0)
    mcmc_amd.hmc_mass_adapted_per_chain(t, st, mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc), n_windows=n_windows, mass_out=mass,
                                        first_step_size=first)
    return draws, nacc, mass, theta


def test_per_chain_adaptation_is_a_chain_of_ordinary_runs():
    """One window: part 0 is a plain run (M = I) whose draws give every chain its masses; the rest of the run is an ordinary
    mi_mcmc_hmc_run with mi_chains.mass_diag = the reported masses, continued through draw0 -- so, with the test above, the oracle's
    arithmetic given the masses."""
    d, C, burn, keep, L, eps0 = 24, 130, 40, 9, 8, 0.05
    prec = synth.ill_conditioned_diag(d, 200.0)
    init = synth.initial_states(C, d, seed=3) / np.sqrt(prec)[None, :]
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DIAG, d, prec=prec)
    draws, nacc, mass, theta = _adapt_per_chain(t, init, d, C, burn, keep, L, eps0, 1)
    d2, n2, m2, _ = _adapt_per_chain(t, init, d, C, burn, keep, L, eps0, 1)
    assert np.array_equal(draws, d2) and np.array_equal(mass, m2) and np.array_equal(nacc, n2)      # reproducible
    assert np.isfinite(mass).all() and (mass > 0).all() and not np.array_equal(mass[:, 0], mass[:, 1])
    b1 = burn // 2
    th = np.ascontiguousarray(init.T.copy())
    dr0 = np.zeros((b1, d, C))
    st0 = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=0, n_keep_draws=b1, n_leap_steps=L, step_size=eps0)
    mcmc_amd.run("hmc", t, st0, mcmc_amd.make_chains(th, C, draws=dr0))
    v = dr0.var(axis=0, ddof=1)                          # the kernel's estimator: two passes, Stan's regularisation
    assert np.allclose(1.0 / ((b1 * v + 5e-3) / (b1 + 5.0)), mass, rtol=1e-9)
    dr1 = np.zeros((keep, d, C)); na1 = np.zeros(C, dtype=np.uint64)
    st1 = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=burn - b1, n_keep_draws=keep, n_leap_steps=L, step_size=eps0)
    mcmc_amd.run("hmc", t, st1, mcmc_amd.make_chains(th, C, draws=dr1, n_accept=na1, draw0=b1, mass_diag=mass))
    assert np.array_equal(dr1, draws) and np.array_equal(na1, nacc) and np.array_equal(th, theta)


def test_per_chain_adaptation_mixes_the_ill_conditioned_target():
    """configs[4]'s target family (precisions 1 .. 1e3): chains with the identity mass crawl along the soft dimensions at the step
    the stiff ones allow (0.03); with its own mass estimate (two windows) every chain runs all dimensions at unit frequency and the
    step of the preconditioned metric (0.25)."""
    d, C, burn, keep, L = 64, 512, 150, 60, 16
    prec = synth.ill_conditioned_diag(d, 1.0e3)
    init = synth.initial_states(C, d, seed=3) / np.sqrt(prec)[None, :]
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DIAG, d, prec=prec)
    draws, nacc, mass, _ = _adapt_per_chain(t, init, d, C, burn, keep, L, 0.25, 2, first=0.03)
    ess_adapted = ess_per_chain(draws[:, ::7, :]).min()
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.03)
    plain, _ = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec)
    ess_plain = ess_per_chain(plain[:, ::7, :]).min()
    print(f"min ESS per chain over {keep} draws: identity mass {ess_plain:.2f}, per-chain mass {ess_adapted:.1f}; accept {nacc.mean() / keep:.2f}; "
          f"median mass / precision {np.median(mass / prec[:, None]):.2f}")
    assert ess_adapted > 3 * ess_plain and nacc.mean() / keep > 0.6
    m2 = (draws[-1] ** 2 * prec[:, None]).mean()
    assert abs(m2 - 1) < 0.1                                        # still the right distribution


def test_per_chain_adaptation_validates_its_arguments():
    d, C = 8, 16
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_ISO, d)
    th = np.ascontiguousarray(synth.initial_states(C, d, seed=1).T.copy())
    for bad in (dict(n_burnin_draws=4, n_keep_draws=2), dict(n_burnin_draws=30, n_keep_draws=2, precond_mat=np.eye(d))):
        with pytest.raises(mcmc_amd.MiMcmcError):
            mcmc_amd.hmc_mass_adapted_per_chain(t, mcmc_amd.default_settings(**bad), mcmc_amd.make_chains(th.copy(), C), n_windows=2)
    with pytest.raises(mcmc_amd.MiMcmcError):
        mcmc_amd.hmc_mass_adapted_per_chain(t, mcmc_amd.default_settings(n_burnin_draws=30, n_keep_draws=2), mcmc_amd.make_chains(th.copy(), C), n_windows=0)


def test_pooled_mass_adaptation_on_the_logistic_regression_target():
    """configs[2]'s target family with badly scaled features (column scales 0.1 .. 10: posterior scales differ by two orders of magnitude).
    hmc with a diagonal precond_mat alone runs on the LDS-staged kernel (its DIAGM instantiation), so the pooled mass adaptation works on
    this target at its own dimensions; the last part is an ordinary hmc run with the reported mass (bit-exact against the oracle)."""
    d, N, C, burn, keep, L = 48, 200, 1024, 60, 40, 8
    X, y = synth.logistic_problem(d, N, seed=5)
    X = X * np.logspace(-1, 1, d)[None, :] * np.sqrt(d)
    t = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=X, y=y)
    init = synth.initial_states(C, d, seed=3) * 0.05
    def run(n_windows, eps):
        theta = np.ascontiguousarray(init.T.copy()); draws = np.zeros((keep, d, C)); nacc = np.zeros(C, dtype=np.uint64)
        st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
        mass = mcmc_amd.hmc_mass_adapted(t, st, mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc), n_windows=n_windows)
        return draws, nacc, mass
    draws, nacc, mass = run(3, 0.25)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<") and mcmc_amd.last_kernel().endswith("true>")
    assert np.isfinite(mass).all() and mass.max() / mass.min() > 50           # it found the scales
    ess_adapted = ess_per_chain(draws).min()
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.02)
    plain, gp = mcmc_amd.hmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    ess_plain = ess_per_chain(plain).min()
    print(f"logistic d={d}: min ESS per chain over {keep} draws: identity mass (eps 0.02) {ess_plain:.2f}, pooled diagonal mass (eps 0.25) {ess_adapted:.1f}; "
          f"accept {nacc.mean() / keep:.2f} / {gp['n_accept'].mean() / keep:.2f}")
    assert ess_adapted > 3 * ess_plain and nacc.mean() / keep > 0.5
    # no window: one ordinary run with the mass from the spread of initial_vals -- the oracle reproduces it
    d0, n0, m0 = run(0, 0.05)
    bs = 16
    s = orc.make_settings(seed=5, n_burnin=burn, n_keep=keep, n_leap=L, step=0.05, W=4, hoist=1, precond=np.diag(m0), blocks=4, block_size=bs)
    o, info = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=bs, eta_chains=2), init[:64], s)
    assert np.array_equal(d0[:, :, :64], o) and np.array_equal(n0[:64], info["n_accept"])
