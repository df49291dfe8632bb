"""GPU: NUTS (iterative wavefront tree) vs the recursive CPU oracle -- bit-exact through the C ABI."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _oracle(kind_orc, d, init, st, prec=None, chain0=0, precond=None):
    C = init.shape[0]
    n_tot = int(st.n_burnin_draws + st.n_keep_draws)
    draws = np.zeros((int(st.n_keep_draws), d, C))
    out = dict(n_accept=np.zeros(C, dtype=np.uint64), n_leap=np.zeros(C, dtype=np.uint64),
               eps=np.zeros(C), depth=np.zeros((n_tot, C), dtype=np.uint32))
    for c in range(C):
        t = orc.TargetSpec(kind_orc, d, prec=prec, W=4)
        s = orc.make_settings(seed=int(st.rng_seed_value), n_burnin=int(st.n_burnin_draws),
                              n_keep=int(st.n_keep_draws), step=float(st.step_size),
                              n_adapt=int(st.n_adapt_draws), delta=float(st.target_accept_rate),
                              max_depth=int(st.max_tree_depth), gamma=float(st.gamma_val),
                              t0=float(st.t0_val), kappa=float(st.kappa_val), W=4, chain_id=chain0 + c, precond=precond)
        dr, info = orc.run_chain(orc.ALGO_NUTS, t, init[c], s, traces=True)
        draws[:, :, c] = dr
        out["n_accept"][c] = info["n_accept"]
        out["n_leap"][c] = info["n_leap"]
        out["eps"][c] = info["eps"]
        out["depth"][:, c] = info["depth"]
    return draws, out


CASES = [
    # kind,   d,   C,  burn, keep, adapt, max_depth, eps_bar0
    ("dense", 8, 16, 5, 20, 15, 10, 1.0),        # SURVEY 8(c) golden shape, adaptation on
    ("iso", 3, 20, 30, 30, 30, 10, 1.0),         # 3-D isotropic Gaussian
    ("dense", 128, 16, 6, 6, 8, 10, 1.0),        # BASELINE config 4 shape (one wave)
    ("dense", 128, 70, 3, 4, 4, 10, 1.0),        # ragged: partial waves, dead lanes
    ("dense", 33, 40, 4, 8, 6, 4, 1.0),          # odd d, shallow trees (max_tree_depth cap hit)
    ("diag", 20, 32, 0, 12, 0, 6, 0.05),         # no adaptation: fixed small step -> deep trees
    ("dense", 16, 64, 2, 10, 12, 1, 1.0),        # max_tree_depth = 1
    ("dense", 100, 150, 3, 5, 5, 10, 1.0),       # three workgroups of the split kernel, the last tile partly live, d short of its row tiles
    ("dense", 128, 24, 0, 4, 0, 7, 0.02),        # fixed small step at d = 128: trees to the cap, every level of the unwind
    ("dense", 65, 16, 2, 3, 0, 0, 1.0),          # max_tree_depth = 0: the while-loop of nuts.cpp:227 never entered
    ("dense", 128, 200, 10, 14, 12, 8, 1.0),     # four workgroups, the adaptation window ends inside the run (nuts_dyn.hpp: every chain in a slot of its own)
    ("dense", 32, 300, 12, 20, 16, 10, 1.0),     # five workgroups of narrow tiles
]


# AUTO / KERNEL_NUTS_MEMO: every doubling on a memoised trajectory (nuts_memo.hpp), THE kernel of the plain case.  KERNEL_NUTS_REG / _SPLIT / _DYN name
# the register-carried kernels of rounds 2-4, retired in round 5: valid hints, ignored -- the default kernel runs.  KERNEL_NUTS_TICK_LOCAL: the
# tick-local asynchronous kernel (what the bounded / preconditioned variants run; it executes every leaf) -- an independent implementation that must
# give the same bits.  (The lock-step first-generation kernel is only in the A/B library: `make prof`.)
# (round 6: the full matrix runs on AUTO, KERNEL_NUTS_MEMO_INTICK and KERNEL_NUTS_TICK_LOCAL; the three retired hints run the SAME kernel as AUTO and
#  are checked for exactly that -- accepted, ignored -- on one case: test_retired_hints_run_the_default_kernel)
# KERNEL_NUTS_MEMO_INTICK (round 6): the memoised tick generating its momenta inside the tick, as until round 5 -- AUTO reads them from the table a
# pre-pass kernel fills (nuts_memo.hpp: nuts_momenta_kernel) and falls back to this form when the table would not fit; same bits
KERNELS = [mcmc_amd.KERNEL_AUTO, mcmc_amd.KERNEL_NUTS_TICK_LOCAL, mcmc_amd.KERNEL_NUTS_MEMO_INTICK]
RETIRED = [mcmc_amd.KERNEL_NUTS_REG, mcmc_amd.KERNEL_NUTS_SPLIT, mcmc_amd.KERNEL_NUTS_DYN]


@pytest.mark.parametrize("hint", KERNELS)
@pytest.mark.parametrize("kind,d,C,burn,keep,adapt,max_depth,eps0", CASES)
def test_nuts_bit_exact_vs_oracle(kind, d, C, burn, keep, adapt, max_depth, eps0, hint):
    init = synth.initial_states(C, d, seed=21)
    prec, k_gpu, k_orc = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    if kind == "dense":
        prec, k_gpu, k_orc = synth.dense_gaussian_precision(d, seed=6), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
    elif kind == "diag":
        prec, k_gpu, k_orc = synth.ill_conditioned_diag(d, 50.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep,
                                   n_adapt_draws=adapt, max_tree_depth=max_depth, step_size=eps0)
    g_draws, g = mcmc_amd.nuts(k_gpu, init, st, prec=prec, chain0=500, kernel_hint=hint)
    memo = hint != mcmc_amd.KERNEL_NUTS_TICK_LOCAL
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel" if memo else "nuts_gauss_async_kernel")
    if memo:        # ... <NT, DIAGM, momenta from the pre-pass table>
        assert mcmc_amd.last_kernel().endswith(", false>" if hint == mcmc_amd.KERNEL_NUTS_MEMO_INTICK else ", true>"), mcmc_amd.last_kernel()
    o_draws, o = _oracle(k_orc, d, init, st, prec=prec, chain0=500)
    assert np.array_equal(g["depth"], o["depth"])            # same trees
    assert np.array_equal(g["n_leap"], o["n_leap"])          # same executed leapfrogs
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"])                # same dual-averaging trajectory
    assert np.array_equal(g_draws, o_draws)
    assert np.linalg.norm(g_draws - o_draws) <= 1e-9 * np.linalg.norm(o_draws)
    if memo:
        # the leapfrogs it really made: what the memoised oracle makes (one per distinct point of a doubling + the step-size search)
        n_exec = []
        for c in range(C):
            t = orc.TargetSpec(k_orc, d, prec=prec, W=4)
            s = orc.make_settings(seed=77, n_burnin=burn, n_keep=keep, step=eps0, n_adapt=adapt, max_depth=max_depth, W=4, chain_id=500 + c)
            n_exec.append(orc.run_chain(orc.ALGO_NUTS_MEMO, t, init[c], s)[1]["n_exec"])
        assert np.array_equal(g["n_exec"], np.array(n_exec, dtype=np.uint64))
        assert (g["n_exec"] <= g["n_leap"]).all()
    else:
        assert np.array_equal(g["n_exec"], g["n_leap"])     # every other kernel executes what it counts


@pytest.mark.parametrize("hint", RETIRED)
def test_retired_hints_run_the_default_kernel(hint):
    d, C = 128, 70
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=21)
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=3, n_keep_draws=4, n_adapt_draws=4)
    a, ga = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=mcmc_amd.KERNEL_AUTO)
    name = mcmc_amd.last_kernel()
    b, gb = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=hint)
    assert mcmc_amd.last_kernel() == name and name.startswith("nuts_gauss_memo_kernel")
    assert np.array_equal(a, b) and np.array_equal(ga["n_exec"], gb["n_exec"])


def test_nuts_sharding_independence_and_statistics():
    d, C = 32, 512
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=2)
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=60, n_keep_draws=40, n_adapt_draws=60)
    full, gf = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    a, _ = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init[:200], st, prec=prec, chain0=0)
    b, _ = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init[200:], st, prec=prec, chain0=200)
    assert np.array_equal(full, np.concatenate([a, b], axis=2))
    last = full[-1]
    cov = last @ last.T / C
    want = np.linalg.inv(prec)
    assert abs(np.trace(cov) / np.trace(want) - 1) < 0.15
    assert (gf["depth"] <= 10).all() and gf["depth"].max() >= 2
    assert 0.3 < gf["n_accept"].mean() / 40 <= 1.0


# ---------------------------------------------------------------- box constraints and diagonal precond_mat
def _bounds(d, seed=0):
    """A mix of the four bounds types of determine_bounds_type.hpp:27-57."""
    rng = np.random.default_rng(seed)
    kind = rng.integers(1, 5, d)
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf)
    ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    return lb, ub


@pytest.mark.parametrize("d,C,burn,keep,bounded,precond", [
    (6, 16, 6, 10, True, False), (128, 32, 3, 4, True, False), (20, 24, 5, 8, False, True), (37, 20, 4, 6, True, True)])
def test_general_nuts_bit_exact_vs_oracle(d, C, burn, keep, bounded, precond):
    prec = synth.dense_gaussian_precision(d, seed=5)
    init = np.clip(synth.initial_states(C, d, seed=14) * 0.3, -1.0, 1.5)     # inside every box
    kw, okw = {}, {}
    if bounded:
        lb, ub = _bounds(d, seed=d)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    if precond:
        M = np.diag(1.0 / np.diag(prec) * np.linspace(0.5, 2.0, d))
        kw.update(precond_mat=M); okw.update(precond=M)
    st = mcmc_amd.default_settings(rng_seed_value=79, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=burn, max_tree_depth=6, **kw)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=79, n_burnin=burn, n_keep=keep, n_adapt=burn, max_depth=6, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=4)
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g["eps"], o["eps"])
    assert np.array_equal(g_draws, o_draws)
    if bounded:
        assert ((g_draws >= lb[None, :, None]) & (g_draws <= ub[None, :, None])).all()   # reported in the constrained space


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.5, 1.5, d))


@pytest.mark.parametrize("d,C,bounded", [(8, 16, False), (37, 24, False), (64, 20, False), (20, 24, True),
                                        (128, 40, False), (100, 20, True)])      # d > 64: INV(M), CHOL(M) from L2 in fragment order
def test_dense_precond_nuts_bit_exact_vs_oracle(d, C, bounded):
    prec = synth.dense_gaussian_precision(d, seed=5)
    M = _spd(d, seed=d)
    init = np.clip(synth.initial_states(C, d, seed=14) * 0.3, -1.0, 1.5)
    kw, okw = dict(precond_mat=M), dict(precond=M)
    if bounded:
        lb, ub = _bounds(d, seed=d)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=80, n_burnin_draws=4, n_keep_draws=6, n_adapt_draws=4, max_tree_depth=6, **kw)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=80, n_burnin=4, n_keep=6, n_adapt=4, max_depth=6, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=4)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"])
    assert np.array_equal(g_draws, o_draws)


@pytest.mark.parametrize("d", [65, 100, 127, 128])
@pytest.mark.parametrize("general", ["bounds", "bounds_diag_precond", "dense_precond"])
@pytest.mark.parametrize("n_adapt", [0, 3])
def test_general_nuts_between_d64_and_d128(d, general, n_adapt):
    """The general kernels -- bounds: the memoised tick with the tile route's policy (nuts_bounded_launch.hip); a dense precond_mat:
    nuts_gauss_async_kernel<8, true, true> -- with dimensions that do not fill the eight row tiles (lanes without a dimension, partial
    exec masks): a fuzz sweep of round 3 caught this shape returning the chain index as step size when three more loop-carried scalars
    had pushed the register allocation over an edge.  Pinned here: draws, leapfrog counts, adapted step sizes against the oracle."""
    C = 5
    prec = synth.dense_gaussian_precision(d, seed=5)
    init = np.clip(synth.initial_states(C, d, seed=7), -1.0, 1.5)
    kw, okw = {}, {}
    rng = np.random.default_rng(d)
    if "bounds" in general:
        lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 1, 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    if "diag_precond" in general:
        M = np.diag(np.linspace(0.5, 2.0, d)); kw.update(precond_mat=M); okw.update(precond=M)
    if "dense_precond" in general:
        A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(np.linspace(0.5, 2.0, d)); kw.update(precond_mat=M); okw.update(precond=M)
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=3, n_keep_draws=4, step_size=0.05, n_adapt_draws=n_adapt, max_tree_depth=4, **kw)
    g_draws, g = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, want_adapt_state=True)
    assert mcmc_amd.last_kernel().startswith("nuts_tile_kernel<built-in Gaussian 8, true>" if "bounds" in general else "nuts_gauss_async_kernel<8, true, true>")
    s = orc.make_settings(seed=3, n_burnin=3, n_keep=4, step=0.05, n_adapt=n_adapt, max_depth=4, W=4, hoist=1, **okw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    assert np.array_equal(g["eps"], o["eps"]) and np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)


def test_memoised_kernel_at_every_launch_shape_gives_the_bits_of_the_tick_local_kernel():
    """nuts_gauss_memo_kernel<8> on 700 / 5 000 / 20 000 chains -- one, two and four waves per workgroup, the last beyond the chip's 16 384 chain
    slots (a persistent grid whose slots take a second chain): all equal to nuts_gauss_async_kernel (which executes every leaf) bit for bit, ragged
    last tiles included, with fewer leapfrogs executed."""
    d = 128
    prec = synth.dense_gaussian_precision(d, seed=6)
    st = mcmc_amd.default_settings(rng_seed_value=11, n_burnin_draws=4, n_keep_draws=3, n_adapt_draws=4, max_tree_depth=6)
    for C in (700, 5000, 20000):
        init = synth.initial_states(C, d, seed=C)
        ref, r = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=3, kernel_hint=mcmc_amd.KERNEL_NUTS_TICK_LOCAL)
        assert mcmc_amd.last_kernel().startswith("nuts_gauss_async_kernel")
        got, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=3)
        assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel<8, false, true>")     # (momenta from the pre-pass table)
        assert np.array_equal(got, ref)
        for k in ("n_accept", "n_leap", "eps", "depth"):
            assert np.array_equal(g[k], r[k]), k
        assert np.array_equal(r["n_exec"], r["n_leap"]) and (g["n_exec"] <= g["n_leap"]).all() and g["n_exec"].sum() < 0.9 * g["n_leap"].sum()
