"""GPU: mi_mcmc_rmhmc_run_callback -- mcmc::rmhmc with HOST callbacks (ref: include/mcmc/rmhmc.hpp, src/rmhmc.cpp:30-287).  The sampler is
the device's literal rmhmc kernel; every target / tensor evaluation is a request to the host (mcmc_amd/csrc/literal.hpp: LitMailbox).
The user callbacks here are the ORACLE's own target and tensor functions, passed as C function pointers: the run must then equal the
oracle's rmhmc bit for bit -- which pins the sampler's arithmetic, the callback order and the mailbox plumbing at once."""
import ctypes as C

import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _ptrs():
    lib = orc.lib()
    return C.cast(lib.orc_target_kernel, C.c_void_p).value, C.cast(lib.orc_target_tensor, C.c_void_p).value


@pytest.mark.parametrize("case", ["logistic_d5", "logistic_d12_bounded", "dense_gaussian_d9", "normal_model"])
def test_rmhmc_with_host_callbacks_equals_the_oracle(case):
    kw, okw = {}, {}
    if case.startswith("logistic"):
        d = 5 if "d5" in case else 12
        X, y = synth.logistic_problem(d, 30, seed=3)
        t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4)
        init = synth.initial_states(1, d, seed=2)[0] * 0.2
        burn, keep, L, eps, nfp = 2, 6, 2, 0.08, 3
        if "bounded" in case:
            lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 1, 2.0, np.inf)
            kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
            init = np.clip(init, -1.0, 1.5)
    elif case.startswith("dense"):
        d = 9
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=synth.dense_gaussian_precision(d, seed=4), W=4)
        init = synth.initial_states(1, d, seed=5)[0]
        burn, keep, L, eps, nfp = 1, 8, 3, 0.05, 2
    else:
        d = 2
        x = 2.0 + 2.0 * np.random.default_rng(1).standard_normal(200)
        t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, d, y=x, W=4)
        init = np.array([1.5, 2.5])
        burn, keep, L, eps, nfp = 3, 10, 2, 0.05, 4
        lb = np.array([-np.inf, 0.01]); ub = np.array([np.inf, np.inf])
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    kern, tens = _ptrs()
    st = mcmc_amd.default_settings(rng_seed_value=21, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps, n_fp_steps=nfp, **kw)
    g_before, v_before = t.c.n_grad_calls, t.c.n_value_calls
    draws, nacc = mcmc_amd.rmhmc_callback(init, kern, C.addressof(t.c), tens, C.addressof(t.c), st)
    n_grad, n_value = t.c.n_grad_calls - g_before, t.c.n_value_calls - v_before
    assert mcmc_amd.last_kernel() == "literal_kernel<4>"
    s = orc.make_settings(seed=21, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, n_fp=nfp, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_RMHMC, t, init[None, :], s)
    assert np.array_equal(draws, o_draws[:, :, 0]) and nacc == int(o["n_accept"][0])
    assert 0 < nacc
    # called exactly where the reference calls them (src/rmhmc.cpp:199-272): n_fp + 1 gradients per leapfrog step, one value per draw + setup
    n_tot = burn + keep
    assert n_grad == n_tot * L * (nfp + 1) and n_value == n_tot + 1


def test_rmhmc_callback_route_validates_its_arguments():
    kern, tens = _ptrs()
    t = orc.TargetSpec(orc.TARGET_ISO, 3, W=4)
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1)
    with pytest.raises(mcmc_amd.MiMcmcError):
        mcmc_amd.rmhmc_callback(np.zeros(3), None, None, tens, C.addressof(t.c), st)
    with pytest.raises(mcmc_amd.MiMcmcError):
        mcmc_amd.rmhmc_callback(np.zeros(65), kern, C.addressof(t.c), tens, C.addressof(t.c), st)


# ---- the same mailbox under the other samplers: a host callback target together with bounds / precond_mat (cov_mat) runs on the literal
# kernel with the callback as its target (round 2 refused these combinations; ref: src/hmc.cpp:57-59,84-95,107-122, src/mala.cpp:152-157,
# src/nuts.cpp:139-154, src/rwmh.cpp:58,119)
@pytest.mark.parametrize("algo", ["hmc", "mala", "nuts", "rwmh"])
@pytest.mark.parametrize("general", ["bounds", "diag_precond", "bounds_dense_precond"])
def test_host_callback_target_with_bounds_and_preconditioner_equals_the_oracle(algo, general):
    d = 7
    X, y = synth.logistic_problem(d, 25, seed=8)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4)
    rng = np.random.default_rng(3)
    init = np.clip(synth.initial_states(1, d, seed=6)[0] * 0.3, -1.0, 1.5)
    kw, okw = {}, {}
    if "bounds" in general:
        lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 1, 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    if "precond" in general:
        M = np.diag(rng.uniform(0.5, 2.0, d))
        if "dense" in general:
            A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + M
        kw.update(precond_mat=M); okw.update(precond=M)
    st = mcmc_amd.default_settings(rng_seed_value=14, n_burnin_draws=3, n_keep_draws=7, n_leap_steps=4, step_size=0.15, n_adapt_draws=3,
                                   max_tree_depth=4, **kw)
    draws, nacc = mcmc_amd.hmc_callback(init, orc.lib().orc_target_kernel, st, target_data=C.addressof(t.c), algo=algo)
    assert mcmc_amd.last_kernel() == "literal_kernel<%d>" % {"hmc": 0, "mala": 1, "nuts": 2, "rwmh": 3}[algo]
    s = orc.make_settings(seed=14, n_burnin=3, n_keep=7, n_leap=4, step=0.15, n_adapt=3, max_depth=4, W=4, **okw)
    o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH}[algo], t, init[None, :], s)
    assert np.array_equal(draws, o_draws[:, :, 0]) and nacc == int(o["n_accept"][0])
    assert nacc > 0 or (algo == "nuts" and np.isfinite(draws).all())      # (a short adapting nuts run may keep its state)
