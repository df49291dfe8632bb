"""GPU: logistic regression with d <= 8 coefficients on the one-lane-per-chain engine (small_targets.hpp: LogisticSmallModel) --
what the LDS-staged MFMA kernel does not implement: mcmc::nuts, vals_bound, precond_mat / cov_mat on MI_TARGET_LOGISTIC.
Bit-exact against the oracle in the LDS kernel's reduction order (W = 4, 4 blocks of 16, 2 eta sub-chains), and -- with an
identity precond_mat, which routes hmc / mala / rwmh here -- bit-identical to that kernel on the same problem."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu
ALGO = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "rwmh": orc.ALGO_RWMH, "nuts": orc.ALGO_NUTS}


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.5, 1.5, d))


def _bounds(d, kind):
    lo, hi = np.full(d, -np.inf), np.full(d, np.inf)
    if kind == "box":
        lo[:], hi[:] = -2.5, 3.0
    elif kind == "mixed":
        lo[0::3] = -3.0; hi[1::3] = 2.5; lo[2::3] = -2.0; hi[2::3] = 2.0
    return lo, hi


def _oracle(algo, d, X, y, init, chain0, precond=None, bounds=None, **kw):
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=16, eta_chains=2)
    okw = dict(kw)
    if precond is not None:
        okw["precond"] = precond
    if bounds is not None:
        okw["lower"], okw["upper"] = bounds
    s = orc.make_settings(W=4, hoist=0, blocks=4, block_size=16, **okw)
    draws = np.zeros((s.n_keep_draws, d, init.shape[0]))
    info = dict(n_accept=[], n_leap=[], eps=[], depth=[])
    for c in range(init.shape[0]):
        s.chain_id = chain0 + c
        dr, i = orc.run_chain(ALGO[algo], t, init[c], s, traces=True)
        draws[:, :, c] = dr
        for k in info:
            info[k].append(i[k])
    return draws, {k: np.array(v) for k, v in info.items()}


CASES = [
    # algo, d, N, C, step, L, burn, keep, precond, bounds
    ("nuts", 5, 40, 16, 1.0, 1, 5, 20, None, None),            # the committed golden shape nuts_logit5
    ("nuts", 8, 100, 70, 1.0, 1, 6, 10, None, None),
    ("nuts", 1, 30, 33, 1.0, 1, 4, 8, None, None),
    ("nuts", 3, 50, 20, 1.0, 1, 5, 10, "dense", "box"),
    ("nuts", 7, 64, 24, 1.0, 1, 4, 8, "diag", None),
    ("hmc", 5, 40, 40, 0.05, 6, 3, 12, None, "box"),
    ("hmc", 8, 60, 33, 0.04, 4, 2, 10, "dense", None),
    ("hmc", 6, 50, 20, 0.04, 3, 2, 10, "dense", "mixed"),
    ("mala", 4, 40, 40, 0.10, 1, 3, 12, None, "mixed"),
    ("mala", 8, 60, 33, 0.08, 1, 2, 10, "dense", None),
    ("mala", 2, 50, 20, 0.08, 1, 2, 10, "dense", "box"),        # dense preconditioner + bounds: INV(eps^2 J M) per draw
    ("rwmh", 5, 40, 40, 0.20, 1, 3, 12, "dense", None),
    ("rwmh", 7, 60, 33, 0.15, 1, 2, 10, None, "box"),
]


@pytest.mark.parametrize("algo,d,N,C,step,L,burn,keep,precond,bounds", CASES)
def test_small_logistic_bit_exact_vs_oracle(algo, d, N, C, step, L, burn, keep, precond, bounds):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.3
    M = None if precond is None else (_spd(d, seed=d) if precond == "dense" else np.diag(np.linspace(0.6, 1.7, d)))
    bd = None if bounds is None else _bounds(d, bounds)
    kw = {}
    if M is not None:
        kw["precond_mat"] = M
    if bd is not None:
        kw.update(vals_bound=1, lower_bounds=bd[0], upper_bounds=bd[1])
    st = mcmc_amd.default_settings(rng_seed_value=321, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=step,
                                   n_adapt_draws=burn, max_tree_depth=7, **kw)
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=11)
    o_draws, o = _oracle(algo, d, X, y, init, 11, precond=M, bounds=bd, seed=321, n_burnin=burn, n_keep=keep, n_leap=L, step=step,
                         n_adapt=burn, max_depth=7)
    assert np.array_equal(g["n_accept"], o["n_accept"].astype(np.uint64))
    assert np.array_equal(g_draws, o_draws)
    if algo == "nuts":
        assert np.array_equal(g["n_leap"], o["n_leap"].astype(np.uint64)) and np.array_equal(g["eps"], o["eps"])
    if bd is not None:
        assert ((g_draws >= bd[0][None, :, None]) & (g_draws <= bd[1][None, :, None])).all()
    assert g["n_accept"].sum() > 0


@pytest.mark.parametrize("algo,step,L", [("hmc", 0.05, 5), ("mala", 0.1, 1), ("rwmh", 0.2, 1)])
@pytest.mark.parametrize("d", [3, 8])
def test_one_lane_engine_and_lds_kernel_give_the_same_bits(algo, step, L, d):
    """An identity precond_mat / cov_mat is arithmetically neutral (dense products with exact zeros and ones) and routes the
    run to the one-lane-per-chain engine; without it the LDS-staged MFMA kernel runs: two independent kernels, same draws."""
    N, C = 80, 50
    X, y = synth.logistic_problem(d, N, seed=6)
    init = synth.initial_states(C, d, seed=5) * 0.2
    base = dict(rng_seed_value=17, n_burnin_draws=3, n_keep_draws=9, n_leap_steps=L, step_size=step)
    a, ga = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, mcmc_amd.default_settings(**base), X=X, y=y)
    b, gb = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, mcmc_amd.default_settings(precond_mat=np.eye(d), **base), X=X, y=y)
    assert np.array_equal(a, b) and np.array_equal(ga["n_accept"], gb["n_accept"])


def test_logistic_nuts_beyond_8_dims_runs_on_the_lds_streamed_kernel_and_on_the_literal_kernel_by_request():
    d, C = 9, 6
    X, y = synth.logistic_problem(d, 30, seed=1)
    init = synth.initial_states(C, d, seed=2) * 0.3
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=2, n_keep_draws=3, n_adapt_draws=2, max_tree_depth=4)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    assert mcmc_amd.last_kernel() == "logit_lds_kernel<1, nuts, 0, false>"     # round 4 (nuts_lds.hpp); round 3: the literal kernel
    l_draws, l = mcmc_amd.nuts(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, kernel_hint=mcmc_amd.KERNEL_LITERAL)
    assert mcmc_amd.last_kernel() == "literal_kernel<2>"
    assert np.array_equal(g_draws, l_draws) and np.array_equal(g["n_leap"], l["n_leap"]) and np.array_equal(g["eps"], l["eps"])
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=16, eta_chains=2)
    s = orc.make_settings(seed=4, n_burnin=2, n_keep=3, n_adapt=2, max_depth=4, step=float(st.step_size), W=4, blocks=4, block_size=16)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s)
    assert np.array_equal(g_draws, o_draws) and np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"])
