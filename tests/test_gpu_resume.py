"""GPU: checkpoint / resume through mi_chains.draw0 -- a run cut in two calls (second call: draw0 = draws done so far,
initial values = final theta of the first) is bit-identical to the uncut run.  SURVEY 8 (f-3)."""
import numpy as np
import pytest

import mcmc_amd
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _cut_vs_whole(algo, kind, d, C, k1, k2, burn, tkw, **skw):
    init = synth.initial_states(C, d, seed=8) * 0.5
    whole = mcmc_amd.default_settings(rng_seed_value=321, n_burnin_draws=burn, n_keep_draws=k1 + k2, **skw)
    w_draws, w = mcmc_amd.sample(algo, kind, init, whole, chain0=9, **tkw)
    first = mcmc_amd.default_settings(rng_seed_value=321, n_burnin_draws=burn, n_keep_draws=k1, **skw)
    a_draws, a = mcmc_amd.sample(algo, kind, init, first, chain0=9, **tkw)
    second = mcmc_amd.default_settings(rng_seed_value=321, n_burnin_draws=0, n_keep_draws=k2, **skw)
    b_draws, b = mcmc_amd.sample(algo, kind, a["theta"].T.copy(), second, chain0=9, draw0=burn + k1,
                                 step_size_in=a["eps"] if algo == "nuts" else None, **tkw)
    assert np.array_equal(np.concatenate([a_draws, b_draws]), w_draws)
    assert np.array_equal(a["n_accept"] + b["n_accept"], w["n_accept"])
    assert np.array_equal(b["theta"], w["theta"])
    return w, a, b


@pytest.mark.parametrize("algo,kind,d", [("hmc", "dense", 128), ("hmc", "iso", 200), ("mala", "dense", 40), ("hmc", "diag", 24)])
def test_gaussian_runs_resume_bit_exactly(algo, kind, d):
    tkw, k = {}, mcmc_amd.TARGET_GAUSS_ISO
    if kind == "dense": tkw, k = dict(prec=synth.dense_gaussian_precision(d, seed=2)), mcmc_amd.TARGET_GAUSS_DENSE
    if kind == "diag": tkw, k = dict(prec=synth.ill_conditioned_diag(d, 10.0)), mcmc_amd.TARGET_GAUSS_DIAG
    _cut_vs_whole(algo, k, d, 37, 5, 6, 3, tkw, n_leap_steps=4, step_size=0.1)


@pytest.mark.parametrize("algo", ["mala", "hmc"])
def test_logistic_runs_resume_bit_exactly(algo):
    d, N = 70, 60
    X, y = synth.logistic_problem(d, N, seed=3)
    _cut_vs_whole(algo, mcmc_amd.TARGET_LOGISTIC, d, 40, 4, 3, 2, dict(X=X, y=y), n_leap_steps=3, step_size=0.05)


def test_nuts_resumes_after_its_adaptation_window():
    d = 32
    tkw = dict(prec=synth.dense_gaussian_precision(d, seed=2))
    w, a, b = _cut_vs_whole("nuts", mcmc_amd.TARGET_GAUSS_DENSE, d, 24, 4, 5, 6, tkw, n_adapt_draws=6, max_tree_depth=6)
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(b["eps"], w["eps"])
    # a continuation inside the adaptation window needs the dual-averaging state of the call before (mi_chains.nuts_adapt_state): refused without
    st = mcmc_amd.default_settings(n_burnin_draws=0, n_keep_draws=2, n_adapt_draws=6)
    with pytest.raises(mcmc_amd.MiMcmcError) as e:
        mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, np.zeros((4, d)), st, draw0=3, step_size_in=np.ones(4), **tkw)
    assert e.value.code == mcmc_amd.MI_ERR_BAD_ARG


@pytest.mark.parametrize("route", ["memo_d32", "memo_diag_mass_d32", "memo_150chains_d32", "memo_150chains_diag_mass_d100", "memo_d12", "ignored_hint_d100", "tick_local_d32", "general_dense_precond_d20", "small_normal_model", "literal_d150", "literal_depth12"])
@pytest.mark.parametrize("cut", [3, 9, 10, 14])
def test_nuts_can_be_cut_anywhere_with_the_dual_averaging_state(route, cut):
    """checkpoint of (theta, eps, h, Philox counter).  n_adapt_draws = 10 of 12 burn-in + 6 kept draws; the run is cut after
    `cut` draws -- inside the adaptation window (3, 9), exactly at its end (10: that draw still runs at the last dual-averaging step), or
    after it (14) -- and continued with step_size + nuts_adapt_state: bit-identical to the uncut run on every nuts kernel."""
    burn, keep, n_adapt, C = 12, 6, 10, 21
    kw, tkw = dict(max_tree_depth=5), {}
    hint = (mcmc_amd.KERNEL_NUTS_SPLIT if route.startswith("ignored") else mcmc_amd.KERNEL_NUTS_TICK_LOCAL if route.startswith("tick_local")
            else mcmc_amd.KERNEL_AUTO)         # (AUTO: nuts_memo.hpp; a retired kernel's hint is ignored; the tick-local kernel is the independent route)
    if route.startswith("memo") or route.startswith("general") or route.startswith("ignored") or route.startswith("tick_local"):
        d = 100 if route.endswith("d100") else 12 if route.endswith("d12") else 20 if route.startswith("general") else 32
        if "150chains" in route: C = 150
        kind = mcmc_amd.TARGET_GAUSS_DENSE; tkw = dict(prec=synth.dense_gaussian_precision(d, seed=2))
        if "diag_mass" in route: kw["precond_mat"] = np.diag(np.linspace(0.5, 2.0, d))
        if "dense_precond" in route:                          # (the general tick-local kernel; bounds are left out on purpose: a checkpoint holds
            A = np.random.default_rng(2).standard_normal((d, d)) / np.sqrt(d)        # theta in the natural space, and transform(inv_transform(.))
            kw["precond_mat"] = A @ A.T + np.diag(np.linspace(0.5, 2.0, d))          # is not the identity in floating point)
    elif route == "small_normal_model":
        d = 2; kind = mcmc_amd.TARGET_NORMAL_MODEL; tkw = dict(y=2.0 + 2.0 * np.random.default_rng(1).standard_normal(100))
    elif route == "literal_d150":
        d = 150; C = 5; kind = mcmc_amd.TARGET_GAUSS_DENSE; tkw = dict(prec=synth.dense_gaussian_precision(d, seed=2))
    else:
        d = 6; C = 5; kind = mcmc_amd.TARGET_GAUSS_ISO; kw["max_tree_depth"] = 12; kw["step_size"] = 0.02
    init = synth.initial_states(C, d, seed=8) * 0.5
    if route == "small_normal_model": init = np.abs(init) + np.array([1.5, 1.5])
    S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=n_adapt, **kw)
    # the uncut run keeps every draw (burn-in included) so that the two halves can be compared row by row
    w_draws, w = mcmc_amd.sample("nuts", kind, init, S(0, burn + keep), chain0=4, want_adapt_state=True, kernel_hint=hint, **tkw)
    a_draws, a = mcmc_amd.sample("nuts", kind, init, S(0, cut), chain0=4, want_adapt_state=True, kernel_hint=hint, **tkw)
    b_draws, b = mcmc_amd.sample("nuts", kind, a["theta"].T.copy(), S(0, burn + keep - cut), chain0=4, draw0=cut, step_size_in=a["eps"],
                                 adapt_state_in=a["adapt_state"], kernel_hint=hint, **tkw)
    if route.startswith("memo") or route.startswith("ignored"):
        assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel<")
    if route.startswith("tick_local"):
        assert mcmc_amd.last_kernel().startswith("nuts_gauss_async_kernel<")
    assert np.array_equal(np.concatenate([a_draws, b_draws]), w_draws)
    assert np.array_equal(b["eps"], w["eps"])
    if cut <= n_adapt:                                  # (after the window the state is no longer read, hence not carried)
        assert np.array_equal(b["adapt_state"], w["adapt_state"])
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(np.concatenate([a["depth"], b["depth"]]), w["depth"])
