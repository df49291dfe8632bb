"""GPU: mcmc::nuts on the LDS-streamed evaluation (mcmc_amd/csrc/nuts_lds.hpp; ref: src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241) --
the logistic-regression target with 8 < d <= 512 and dense Gaussians with 128 < d <= 512: the per-chain tree state machine on vectors
split over the four waves of a chain tile, every dot product over dimensions an exchange.  Bit for bit against the oracle (blocked
reduction orders: four dimension quarters), against the literal kernel (same library, one workgroup per chain) at sizes the oracle
does not finish in seconds, across a checkpoint, and in the non-finite regime (flagged chains are replayed literally)."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _bs(kind, d):
    if kind == "logistic":
        return 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    return 48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128


def _problem(kind, d, n_rows, seed):
    if kind == "logistic":
        X, y = synth.logistic_problem(d, n_rows, seed=seed)
        tkw = dict(X=X, y=y)
        spec = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=_bs(kind, d), eta_chains=2)
        return mcmc_amd.TARGET_LOGISTIC, tkw, spec
    prec = synth.dense_gaussian_precision(d, seed=seed)
    return mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=prec), orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, blocks=4, block_size=_bs(kind, d))


def _same(g_draws, g, o_draws, o, depth=True):
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True)
    if depth:
        assert np.array_equal(g["depth"], o["depth"])


# every instantiation: logistic NTQ = 1, 2, 4, 8; dense NTQ = 3, 4, 6, 8.  C = 37: a full workgroup and a ragged one; C = 5: one tile, ragged
CASES = [("logistic", 20, 37, 5, 6), ("logistic", 100, 16, 37, 5), ("logistic", 200, 50, 5, 5), ("logistic", 512, 37, 37, 4),
         ("dense", 160, 0, 37, 5), ("dense", 256, 0, 5, 6), ("dense", 300, 0, 5, 5), ("dense", 512, 0, 37, 4)]


@pytest.mark.parametrize("kind,d,n_rows,C,depth", CASES)
def test_lds_nuts_matches_the_oracle(kind, d, n_rows, C, depth):
    tk, tkw, spec = _problem(kind, d, n_rows, seed=d)
    init = synth.initial_states(C, d, seed=d + 1) * (0.1 if kind == "logistic" else 0.5)
    eps = 0.05 if kind == "logistic" else 0.1
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=3, n_keep_draws=3, n_adapt_draws=4, max_tree_depth=depth, step_size=eps)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=3, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<") and "nuts" in mcmc_amd.last_kernel()
    s = orc.make_settings(seed=7, n_burnin=3, n_keep=3, n_adapt=4, max_depth=depth, step=eps, W=4, blocks=4, block_size=_bs(kind, d))
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=3)
    assert o["n_leap"].max() > 2 ** (depth - 1), "the case is meant to grow trees of several levels"
    _same(g_draws, g, o_draws, o, depth=False)
    # the leapfrogs it really made (round 6: every doubling on a memoised trajectory): what the memoised oracle makes -- one per distinct point of a
    # doubling + the step-size search -- fewer than the reference counts
    n_exec = []
    for c in range(C):
        sc = orc.make_settings(seed=7, n_burnin=3, n_keep=3, n_adapt=4, max_depth=depth, step=eps, W=4, blocks=4, block_size=_bs(kind, d), chain_id=3 + c)
        n_exec.append(orc.run_chain(orc.ALGO_NUTS_MEMO, spec, init[c], sc)[1]["n_exec"])
    assert np.array_equal(g["n_exec"], np.array(n_exec, dtype=np.uint64))
    assert (g["n_exec"] <= g["n_leap"]).all() and g["n_exec"].sum() < g["n_leap"].sum()


@pytest.mark.parametrize("kind,d,n_rows,C", [("logistic", 100, 20, 37), ("logistic", 512, 24, 5), ("dense", 160, 0, 5), ("dense", 384, 0, 37)])
def test_lds_nuts_with_a_diagonal_precond_mat_matches_the_oracle(kind, d, n_rows, C):
    """ref: src/nuts.cpp:57-59,168,202-204 -- p = sqrt(M) z, K = p.(INV(M) p) / 2, theta += e INV(M) p with a diagonal M"""
    tk, tkw, spec = _problem(kind, d, n_rows, seed=d + 9)
    init = synth.initial_states(C, d, seed=d + 3) * (0.1 if kind == "logistic" else 0.5)
    M = np.diag(np.random.default_rng(d).uniform(0.4, 2.5, d))
    st = mcmc_amd.default_settings(rng_seed_value=21, n_burnin_draws=3, n_keep_draws=3, n_adapt_draws=4, max_tree_depth=5, step_size=0.05, precond_mat=M)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=2, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<") and mcmc_amd.last_kernel().endswith("true>")
    s = orc.make_settings(seed=21, n_burnin=3, n_keep=3, n_adapt=4, max_depth=5, step=0.05, W=4, blocks=4, block_size=_bs(kind, d), precond=M)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=2)
    assert o["n_leap"].max() > 16
    _same(g_draws, g, o_draws, o, depth=False)
    # the non-finite regime with the tables: flagged, replayed literally with the same M
    init[1] = 1e200
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=2, **tkw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=2)
    _same(g_draws, g, o_draws, o, depth=False)


@pytest.mark.parametrize("kind,d,n_rows,C", [("logistic", 512, 64, 100), ("logistic", 48, 200, 70), ("dense", 512, 0, 70), ("dense", 192, 0, 100)])
def test_lds_nuts_matches_the_literal_kernel_on_longer_runs(kind, d, n_rows, C):
    """deep trees (up to 2^8 leaves), many draws, the adaptation window inside the run, several workgroups that finish at different times"""
    tk, tkw, _ = _problem(kind, d, n_rows, seed=d + 5)
    init = synth.initial_states(C, d, seed=d + 2) * (0.1 if kind == "logistic" else 0.5)
    st = mcmc_amd.default_settings(rng_seed_value=11, n_burnin_draws=8, n_keep_draws=6, n_adapt_draws=6, max_tree_depth=8, step_size=0.05)
    a_draws, a = mcmc_amd.sample("nuts", tk, init, st, chain0=9, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    b_draws, b = mcmc_amd.sample("nuts", tk, init, st, chain0=9, kernel_hint=mcmc_amd.KERNEL_LITERAL, **tkw)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<")
    assert a["depth"].max() >= 5
    _same(a_draws, a, b_draws, b)
    assert np.array_equal(a["theta"], b["theta"])


@pytest.mark.parametrize("kind,d,n_rows,C", [("logistic", 512, 64, 48), ("logistic", 48, 200, 70), ("dense", 512, 0, 40), ("dense", 192, 0, 64)])
def test_lds_nuts_matches_the_oracle_on_longer_runs(kind, d, n_rows, C):
    """The same long runs against the ORACLE (the comparison with literal_kernel<2> above is the engine against itself):
    deep trees (up to 2^8 leaves), 14 draws, the adaptation window inside the run, chains that finish at different times -- draws, accepts,
    reference leapfrog counts, step sizes bit for bit.  (The oracle runs its chains on all host cores; about 10 s.)"""
    tk, tkw, spec = _problem(kind, d, n_rows, seed=d + 5)
    init = synth.initial_states(C, d, seed=d + 2) * (0.1 if kind == "logistic" else 0.5)
    st = mcmc_amd.default_settings(rng_seed_value=11, n_burnin_draws=8, n_keep_draws=6, n_adapt_draws=6, max_tree_depth=8, step_size=0.05)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=9, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    s = orc.make_settings(seed=11, n_burnin=8, n_keep=6, n_adapt=6, max_depth=8, step=0.05, W=4, blocks=4, block_size=_bs(kind, d))
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=9)
    assert g["depth"].max() >= 5 and o["n_leap"].max() > 200
    _same(g_draws, g, o_draws, o, depth=False)


@pytest.mark.parametrize("kind,d,n_rows", [("logistic", 100, 40), ("dense", 200, 0)])
@pytest.mark.parametrize("cut", [3, 10, 14])
def test_lds_nuts_can_be_cut_anywhere(kind, d, n_rows, cut):
    """(theta, eps, dual-averaging state, Philox counter) is a checkpoint on this kernel too"""
    burn, keep, n_adapt, C = 12, 6, 10, 37
    tk, tkw, _ = _problem(kind, d, n_rows, seed=3)
    init = synth.initial_states(C, d, seed=8) * (0.1 if kind == "logistic" else 0.5)
    S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=n_adapt, max_tree_depth=5)
    w_draws, w = mcmc_amd.sample("nuts", tk, init, S(0, burn + keep), chain0=4, want_adapt_state=True, **tkw)
    a_draws, a = mcmc_amd.sample("nuts", tk, init, S(0, cut), chain0=4, want_adapt_state=True, **tkw)
    b_draws, b = mcmc_amd.sample("nuts", tk, a["theta"].T.copy(), S(0, burn + keep - cut), chain0=4, draw0=cut, step_size_in=a["eps"],
                                 adapt_state_in=a["adapt_state"], **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    assert np.array_equal(np.concatenate([a_draws, b_draws]), w_draws)
    assert np.array_equal(b["eps"], w["eps"])
    if cut <= n_adapt:
        assert np.array_equal(b["adapt_state"], w["adapt_state"])
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(np.concatenate([a["depth"], b["depth"]]), w["depth"])


@pytest.mark.parametrize("kind,d,n_rows", [("logistic", 40, 30), ("dense", 160, 0)])
def test_lds_nuts_non_finite_chains_are_replayed_literally(kind, d, n_rows):
    """chains that start where the step-size search or a tree overflows are flagged by the tiled kernel and replayed by literal_kernel<2>:
    the oracle's bits for every chain, the tame ones next to them untouched"""
    C = 37
    tk, tkw, spec = _problem(kind, d, n_rows, seed=12)
    init = synth.initial_states(C, d, seed=5) * (0.1 if kind == "logistic" else 0.5)
    init[3] *= 1e160; init[20] = 1e308; init[33, 0] = np.inf
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=2, n_keep_draws=3, n_adapt_draws=3, max_tree_depth=4, step_size=0.1)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    s = orc.make_settings(seed=5, n_burnin=2, n_keep=3, n_adapt=3, max_depth=4, step=0.1, W=4, blocks=4, block_size=_bs(kind, d))
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s)
    _same(g_draws, g, o_draws, o, depth=False)


@pytest.mark.parametrize("kind,d,n_rows", [("logistic", 300, 24), ("dense", 192, 0)])
def test_lds_nuts_results_do_not_depend_on_sharding_or_slot(kind, d, n_rows):
    """chains shard by GLOBAL chain id.  One call over 100 chains against three calls over [0, 37), [37, 69), [69, 100) with
    chain0 set -- a chain lands in a different slot of a different workgroup each time (nuts_lds.hpp hands chains to slots dynamically);
    its draws, step size, tree depths and leapfrog count are the same bits."""
    C = 100
    tk, tkw, _ = _problem(kind, d, n_rows, seed=31)
    init = synth.initial_states(C, d, seed=13) * (0.1 if kind == "logistic" else 0.5)
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=5, n_keep_draws=4, n_adapt_draws=5, max_tree_depth=6, step_size=0.05)
    w_draws, w = mcmc_amd.sample("nuts", tk, init, st, chain0=1000, **tkw)
    for lo, hi in ((0, 37), (37, 69), (69, 100)):
        p_draws, p = mcmc_amd.sample("nuts", tk, init[lo:hi], st, chain0=1000 + lo, **tkw)
        assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
        assert np.array_equal(p_draws, w_draws[:, :, lo:hi])
        for k in ("n_accept", "n_leap", "eps"):
            assert np.array_equal(p[k], w[k][lo:hi])
        assert np.array_equal(p["depth"], w["depth"][:, lo:hi]) and np.array_equal(p["theta"], w["theta"][:, lo:hi])


def test_lds_nuts_refusals_and_fallbacks_are_the_documented_ones():
    d, C = 160, 4
    prec = synth.dense_gaussian_precision(d, seed=1)
    init = synth.initial_states(C, d, seed=1) * 0.5
    # a dense preconditioner WITH bounds / trees deeper than 10 / max_tree_depth = 0: the literal kernel, not a refusal (bounds alone:
    # tests/test_gpu_lds_bounds.py; a dense preconditioner alone: the streamed kernel since round 6, below)
    A = np.random.default_rng(3).standard_normal((d, d)) / np.sqrt(d)
    lb = np.full(d, -np.inf); ub = np.full(d, np.inf); lb[0] = -50.0; ub[1] = 50.0
    for kw in (dict(precond_mat=A @ A.T + np.eye(d), vals_bound=1, lower_bounds=lb, upper_bounds=ub), dict(max_tree_depth=11), dict(max_tree_depth=0)):
        st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=1, n_keep_draws=1, n_adapt_draws=1, step_size=0.1, **{"max_tree_depth": 3, **kw})
        mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
        assert mcmc_amd.last_kernel().startswith("literal_kernel<")


# ---- round 6: a DENSE precond_mat on the same kernels (nuts_lds.hpp: DENSEM; ref: src/nuts.cpp:57-59 inv_precond_matrix = INV(M), sqrt_precond_matrix =
# CHOL_LOWER(M); :168,202 p = L z; :148 theta += e (Minv p); nuts.ipp:51,66,140 and nuts.cpp:204 K = p.(Minv p) / 2; the U-turn dots are plain): both
# matrices streamed through LDS like the target's, three products per leaf; the literal kernel served this case until round 5
def _dense_m(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))


DM_CASES = [("logistic", 20, 37, 37, 5), ("logistic", 100, 16, 5, 5), ("logistic", 200, 50, 37, 4), ("logistic", 512, 37, 5, 4),
            ("dense", 160, 0, 37, 5), ("dense", 256, 0, 37, 5), ("dense", 300, 0, 5, 4), ("dense", 512, 0, 37, 3)]


@pytest.mark.parametrize("kind,d,n_rows,C,depth", DM_CASES)
def test_lds_nuts_with_a_dense_precond_mat_matches_the_oracle(kind, d, n_rows, C, depth):
    tk, tkw, spec = _problem(kind, d, n_rows, seed=d)
    M = _dense_m(d, d + 11)
    init = synth.initial_states(C, d, seed=d + 1) * (0.1 if kind == "logistic" else 0.5)
    if C > 10:
        init[5] *= 1e200; init[9, 3] = np.inf              # two chains leave the finite regime: flagged, replayed literally with the same matrices
    eps = 0.05 if kind == "logistic" else 0.1
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=3, n_keep_draws=3, n_adapt_draws=4, max_tree_depth=depth, step_size=eps, precond_mat=M)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=3, **tkw)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and "nuts" in kern and kern.endswith("false, false, true>"), kern
    s = orc.make_settings(seed=7, n_burnin=3, n_keep=3, n_adapt=4, max_depth=depth, step=eps, W=4, hoist=1, precond=M, blocks=4, block_size=_bs(kind, d))
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=3)
    assert o["n_leap"].max() > 2 ** (depth - 2), "the case is meant to grow trees of several levels"
    _same(g_draws, g, o_draws, o, depth=False)


@pytest.mark.parametrize("kind,d,n_rows,C", [("logistic", 300, 40, 70), ("dense", 256, 0, 100)])
def test_lds_nuts_with_a_dense_precond_mat_matches_the_literal_kernel_on_longer_runs(kind, d, n_rows, C):
    """more chains than slots of two workgroups' tiles are not needed here (the grid cap test covers recycling): a longer run with the adaptation
    window ending inside it, against literal_kernel<2> of the same library, and cut in two through mi_chains.draw0"""
    tk, tkw, _ = _problem(kind, d, n_rows, seed=d + 2)
    M = _dense_m(d, d + 13)
    init = synth.initial_states(C, d, seed=d + 3) * (0.1 if kind == "logistic" else 0.5)
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=6, n_keep_draws=6, n_adapt_draws=8, max_tree_depth=6, step_size=0.08, precond_mat=M)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, **tkw)
    assert mcmc_amd.last_kernel().endswith("false, false, true>")
    l_draws, l = mcmc_amd.sample("nuts", tk, init, st, kernel_hint=mcmc_amd.KERNEL_LITERAL, **tkw)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<2>")
    _same(g_draws, g, l_draws, l, depth=True)
