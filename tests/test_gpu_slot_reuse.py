"""GPU: the DYNAMIC chain hand-out of the persistent-grid NUTS kernels (nuts_memo.hpp, nuts_lds.hpp) exercised at small sizes
(ADVICE r4, medium): with the test-only grid cap (mi_mcmc_test_set_grid_cap, mi_mcmc_probes.h) the grid is ONE or TWO workgroups, so a few
hundred chains run the global counter, slots that take their second ... fifth chain with every per-chain state reset, retire-on-leave and
INIT / SEARCH in a recycled slot -- what otherwise only runs beyond 16 384 chains.  Bit for bit against the oracle: parity, a chain that goes
non-finite (retired at once, replayed), and a run cut inside its adaptation window."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu

DYN_HINTS = ["KERNEL_NUTS_MEMO", "KERNEL_NUTS_MEMO_INTICK"]      # momenta from the pre-pass table (AUTO) / generated inside the tick


@pytest.fixture
def grid_cap():
    def set_cap(n):
        mcmc_amd.test_set_grid_cap(n)
    yield set_cap
    mcmc_amd.test_set_grid_cap(0)


def _oracle(d, init, st, prec, chain0, precond=None):
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=int(st.rng_seed_value), n_burnin=int(st.n_burnin_draws), n_keep=int(st.n_keep_draws), step=float(st.step_size),
                          n_adapt=int(st.n_adapt_draws), max_depth=int(st.max_tree_depth), W=4, precond=precond)
    return orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=chain0)


@pytest.mark.parametrize("hint", DYN_HINTS)
@pytest.mark.parametrize("d,C,cap,burn,keep,adapt,depth", [
    (128, 200, 1, 8, 6, 8, 8),       # 64 slots, 200 chains: every slot takes 3-4 chains; the adaptation window covers the burn-in
    (128, 333, 2, 3, 4, 4, 10),      # two workgroups racing for the counter; ragged last hand-out (333 - 128 = 205 chains from the counter)
    (32, 300, 1, 12, 10, 16, 10),    # narrow tiles; the window ends inside the kept draws
    (100, 70, 1, 0, 5, 0, 7),        # 6 chains from the counter only; no adaptation, fixed small step: deep trees
])
def test_recycled_slots_match_the_oracle(hint, d, C, cap, burn, keep, adapt, depth, grid_cap):
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=23)
    st = mcmc_amd.default_settings(rng_seed_value=31, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=adapt, max_tree_depth=depth,
                                   step_size=1.0 if adapt else 0.05)
    grid_cap(cap)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=1000, kernel_hint=getattr(mcmc_amd, hint))
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel")
    o_draws, o = _oracle(d, init, st, prec, 1000)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"])
    assert np.array_equal(g_draws, o_draws)


@pytest.mark.parametrize("hint", DYN_HINTS)
def test_recycled_slots_with_a_diagonal_precond_mat(hint, grid_cap):
    d, C = 100, 180
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=5)
    M = np.diag(np.linspace(0.5, 2.0, d))
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=5, n_keep_draws=5, n_adapt_draws=5, max_tree_depth=7, precond_mat=M)
    grid_cap(1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=7, kernel_hint=getattr(mcmc_amd, hint))
    o_draws, o = _oracle(d, init, st, prec, 7, precond=M)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"]) and np.array_equal(g_draws, o_draws)


def test_recycled_slots_of_the_bounded_tick_on_the_tile_policy(grid_cap):
    """round 6: nuts with vals_bound runs the memoised tick on the tile policy over a PERSISTENT grid too (nuts_bounded_launch.hip:
    one workgroup per CU at most, the workspace sized by the grid) -- one workgroup here, every slot takes its second to fourth chain, one chain
    starts non-finite (the tile policy applies the reference's NaN rules itself: no replay)."""
    d, C = 100, 230
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = np.clip(synth.initial_states(C, d, seed=29) * 0.5, -1.0, 1.5)
    init[140, 7] = np.inf         # a chain from the counter
    lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 0, 2.0, np.inf)
    st = mcmc_amd.default_settings(rng_seed_value=41, n_burnin_draws=5, n_keep_draws=5, n_adapt_draws=5, max_tree_depth=6,
                                   vals_bound=1, lower_bounds=lb, upper_bounds=ub)
    grid_cap(1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11)
    assert mcmc_amd.last_kernel().startswith("nuts_tile_kernel<built-in Gaussian")
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=41, n_burnin=5, n_keep=5, step=1.0, n_adapt=5, max_depth=6, W=4, lower=lb, upper=ub)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=11)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True) and np.array_equal(g_draws, o_draws, equal_nan=True)
    assert (g["n_exec"] <= g["n_leap"]).all()


@pytest.mark.parametrize("d,C,cap", [(128, 200, 1), (48, 330, 2)])
def test_runs_with_a_diagonal_precond_mat_are_cut_into_pieces_too(d, C, cap, grid_cap):
    """nuts_gauss_memo_kernel<., true, false> (momenta generated in the tick; d = 128: the level-loop walk): 19 draws in pieces of 5, a chain started non-finite (flagged,
    replayed by the general variant with the same tables); against the oracle and against the uncut run"""
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=5)
    init[77] *= 1e200
    M = np.diag(np.linspace(0.5, 2.0, d))
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=10, n_keep_draws=9, n_adapt_draws=12, max_tree_depth=6, precond_mat=M)
    grid_cap(cap)
    g_draws, g = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=7, want_adapt_state=True)
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel") and ", true," in mcmc_amd.last_kernel()
    o_draws, o = _oracle(d, init, st, prec, 7, precond=M)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True) and np.array_equal(g_draws, o_draws, equal_nan=True)
    grid_cap(0)
    u_draws, u = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=7, want_adapt_state=True)
    assert np.array_equal(u_draws, g_draws, equal_nan=True)
    for k in ("adapt_state", "eps", "theta", "n_leap", "n_exec", "n_accept", "depth"):
        assert np.array_equal(u[k], g[k], equal_nan=True), k


@pytest.mark.parametrize("mass", [False, True])
def test_the_bounded_tick_on_the_tile_policy_is_cut_into_pieces_too(mass, grid_cap):
    """more chains than chain slots and 8+ draws: the runs of nuts with vals_bound are cut into pieces as the plain kernel's are (nuts_bounded_launch.hip:
    memo_setup_pieces); the hand-over carries theta in the TRANSFORMED space.  Against the oracle, and the cut run exports what the uncut run exports."""
    d, C = 100, 230
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = np.clip(synth.initial_states(C, d, seed=29) * 0.5, -1.0, 1.5)
    init[140, 7] = np.inf
    lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 0, 2.0, np.inf)
    M = np.diag(np.random.default_rng(3).uniform(0.4, 2.5, d)) if mass else None
    kw = dict(precond_mat=M) if mass else {}
    st = mcmc_amd.default_settings(rng_seed_value=41, n_burnin_draws=10, n_keep_draws=9, n_adapt_draws=12, max_tree_depth=5,
                                   vals_bound=1, lower_bounds=lb, upper_bounds=ub, **kw)
    grid_cap(1)
    g_draws, g = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11, want_adapt_state=True)
    assert mcmc_amd.last_kernel().startswith("nuts_tile_kernel<built-in Gaussian")
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=41, n_burnin=10, n_keep=9, step=1.0, n_adapt=12, max_depth=5, W=4, lower=lb, upper=ub, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=11)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True) and np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True)
    grid_cap(0)
    u_draws, u = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=11, want_adapt_state=True)
    assert np.array_equal(u_draws, g_draws, equal_nan=True)
    for k in ("adapt_state", "eps", "theta", "n_leap", "n_exec", "n_accept", "depth"):
        assert np.array_equal(u[k], g[k], equal_nan=True), k


@pytest.mark.parametrize("hint", DYN_HINTS)
def test_a_flagged_chain_leaves_a_recycled_slot_and_is_replayed(hint, grid_cap):
    d, C = 128, 200
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=19)
    init[2] *= 1.0e300            # in its own first slot
    init[90, 3] = np.inf          # a chain from the counter (second occupant of some slot)
    init[150, 100] = np.nan       # third occupant
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=6, n_keep_draws=5, n_adapt_draws=6, max_tree_depth=6, step_size=0.1)
    grid_cap(1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=getattr(mcmc_amd, hint))
    o_draws, o = _oracle(d, init, st, prec, 0)
    assert np.isnan(o_draws[:, :, [2, 90, 150]]).any()
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["n_accept"], o["n_accept"])


@pytest.mark.parametrize("hint", DYN_HINTS)
@pytest.mark.parametrize("cut", [4, 10])
def test_a_run_on_recycled_slots_can_be_cut_inside_the_window(hint, cut, grid_cap):
    d, C, burn, keep, n_adapt = 64, 150, 12, 6, 10
    prec = synth.dense_gaussian_precision(d, seed=2)
    init = synth.initial_states(C, d, seed=8) * 0.5
    S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=n_adapt, max_tree_depth=6)
    grid_cap(1)
    kw = dict(prec=prec, kernel_hint=getattr(mcmc_amd, hint))
    whole, w = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, S(0, burn + keep), want_adapt_state=True, **kw)
    a_draws, a = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, S(0, cut), want_adapt_state=True, **kw)
    b_draws, b = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, a["theta"].T, S(0, burn + keep - cut), draw0=cut, step_size_in=a["eps"],
                                 adapt_state_in=a["adapt_state"], **kw)
    assert np.array_equal(np.concatenate([a_draws, b_draws], axis=0), whole)
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(b["eps"], w["eps"])


@pytest.mark.parametrize("kind,d,n_rows,C", [("logistic", 100, 16, 150), ("dense", 160, 0, 100)])
def test_lds_nuts_on_recycled_slots_matches_the_oracle(kind, d, n_rows, C, grid_cap):
    """nuts_lds.hpp hands chains out the same way (32 slots per workgroup): one workgroup, 100-150 chains"""
    bs = (32 if d <= 128 else 64) if kind == "logistic" else 48
    if kind == "logistic":
        X, y = synth.logistic_problem(d, n_rows, seed=d)
        tk, tkw = mcmc_amd.TARGET_LOGISTIC, dict(X=X, y=y)
        spec = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=bs, eta_chains=2)
    else:
        prec = synth.dense_gaussian_precision(d, seed=d)
        tk, tkw = mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=prec)
        spec = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, blocks=4, block_size=bs)
    init = synth.initial_states(C, d, seed=d + 1) * (0.1 if kind == "logistic" else 0.5)
    init[40] = 1e200                                  # one chain of the second round goes non-finite: flagged, replayed literally
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=3, n_keep_draws=3, n_adapt_draws=4, max_tree_depth=5, step_size=0.05)
    grid_cap(1)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=3, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    s = orc.make_settings(seed=7, n_burnin=3, n_keep=3, n_adapt=4, max_depth=5, step=0.05, W=4, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=3)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["eps"], o["eps"], equal_nan=True)


# ---- round 6: runs cut into PIECES (nuts_launch.hip: MI_MEMO_PIECES work items per chain when there are more chains than chain slots and 8+ draws; a piece that is
# not the first continues its chain in whatever slot is free, exactly as a continuation call does -- nuts_memo_core.hpp, SPLIT).  Same draws, counts and step
# sizes as the oracle's uninterrupted chains: piece boundaries inside the adaptation window, at its end and in the kept draws; draw counts that are no multiple of
# the piece length; chains that go non-finite in their first and in a later piece (PQ_GONE in every later queue); a run that is itself a continuation.
@pytest.mark.parametrize("hint", DYN_HINTS)
@pytest.mark.parametrize("d,C,cap,burn,keep,adapt,depth", [
    (128, 200, 1, 10, 9, 10, 7),     # 19 draws: pieces of 5, 5, 5, 4; the window ends at a piece boundary
    (64, 333, 2, 0, 17, 0, 6),       # two workgroups: pieces migrate between them; no adaptation
    (16, 150, 1, 20, 20, 33, 5),     # pieces of 10; the window ends inside the last piece but one
    (128, 700, 2, 9, 8, 7, 5),       # 5-6 chains per slot
])
def test_runs_cut_into_pieces_match_the_oracle(hint, d, C, cap, burn, keep, adapt, depth, grid_cap):
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = synth.initial_states(C, d, seed=23)
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=adapt, max_tree_depth=depth,
                                   step_size=1.0 if adapt else 0.05)
    grid_cap(cap)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=500, kernel_hint=getattr(mcmc_amd, hint))
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel")
    o_draws, o = _oracle(d, init, st, prec, 500)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["eps"], o["eps"])
    assert np.array_equal(g_draws, o_draws)
    assert (g["n_exec"] <= g["n_leap"]).all() and (g["n_exec"] > 0).all()


@pytest.mark.parametrize("hint", DYN_HINTS)
def test_chains_flagged_in_any_piece_are_replayed(hint, grid_cap):
    d, C = 128, 260
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=19)
    init[2] *= 1.0e300            # flagged at its first evaluation (piece 0, its own slot)
    init[90, 3] = np.inf          # a chain from the counter
    init[150, 100] = np.nan
    init[201] *= 1.0e150          # overflows its energies
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=10, n_keep_draws=10, n_adapt_draws=10, max_tree_depth=6, step_size=0.1)
    grid_cap(1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=getattr(mcmc_amd, hint))
    o_draws, o = _oracle(d, init, st, prec, 0)
    assert np.isnan(o_draws[:, :, [2, 90, 150]]).any()
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["n_accept"], o["n_accept"])


@pytest.mark.parametrize("hint", DYN_HINTS)
def test_a_chain_flagged_in_a_later_piece_is_replayed_from_its_initial_values(hint, grid_cap):
    """a flat target and a tiny gamma_val: dual averaging drives the step size past 1e150 -- the kinetic energy overflows -- after draw 3, 6 or 12 of 20, i.e. in
    piece 0, 1 or 2 of the run.  The replay starts from the chain's INITIAL values, which the earlier pieces have overwritten in mi_chains.theta: the launcher
    keeps a copy (nuts_launch.hip: restore_flagged_theta_kernel)"""
    d, C = 16, 150
    prec = synth.dense_gaussian_precision(d) * 1.0e-100
    init = synth.initial_states(C, d, seed=19)
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=0, n_keep_draws=20, n_adapt_draws=20, max_tree_depth=5, step_size=0.1, gamma_val=2.3e-4)
    grid_cap(1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=getattr(mcmc_amd, hint))
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel")
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=6, n_burnin=0, n_keep=20, step=0.1, n_adapt=20, max_depth=5, W=4, gamma=2.3e-4)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=0)
    assert np.isinf(o["eps"]).sum() > C // 2             # (the regime was reached -- by most chains after draw 12)
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g["theta"], o_draws[-1], equal_nan=True)


def test_a_continuation_call_is_cut_into_pieces_too(grid_cap):
    d, C, n_adapt = 64, 200, 14
    prec = synth.dense_gaussian_precision(d, seed=2)
    init = synth.initial_states(C, d, seed=8) * 0.5
    S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=n_adapt, max_tree_depth=6)
    grid_cap(1)
    kw = dict(prec=prec)
    whole, w = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, S(0, 30), want_adapt_state=True, **kw)
    a_draws, a = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init, S(0, 6), want_adapt_state=True, **kw)          # (6 draws: in one piece)
    b_draws, b = mcmc_amd.sample("nuts", mcmc_amd.TARGET_GAUSS_DENSE, a["theta"].T, S(0, 24), draw0=6, step_size_in=a["eps"],
                                 adapt_state_in=a["adapt_state"], want_adapt_state=True, **kw)                                 # (24 draws from draw 6 on: pieces of 6)
    assert np.array_equal(np.concatenate([a_draws, b_draws], axis=0), whole)
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(b["eps"], w["eps"]) and np.array_equal(b["adapt_state"], w["adapt_state"])
    o_draws, o = _oracle(d, init, S(0, 30), prec, 0)
    assert np.array_equal(whole, o_draws) and np.array_equal(w["eps"], o["eps"])


def test_pieces_without_the_optional_outputs(grid_cap):
    """the hand-over goes through step sizes, counters and the dual-averaging state: the launcher provides stand-ins for the ones the caller did not ask for"""
    import torch
    d, C = 64, 200
    prec = synth.dense_gaussian_precision(d, seed=4)
    init = synth.initial_states(C, d, seed=3) * 0.5
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=10, n_keep_draws=10, n_adapt_draws=10, max_tree_depth=6)
    grid_cap(1)
    full_draws, full = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    dev = torch.device("cuda", 0)
    theta = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    draws = torch.empty((10, d, C), dtype=torch.float64, device=dev)
    tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(prec).to(dev), mem=mcmc_amd.MEM_DEVICE)
    ch = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE)          # theta and draws only
    mcmc_amd.run("nuts", tgt, st, ch)
    torch.cuda.synchronize()
    assert np.array_equal(draws.cpu().numpy(), full_draws) and np.array_equal(theta.cpu().numpy(), full["theta"])


# ---- the same cut on nuts_lds.hpp (logistic_nuts_impl.hpp: MI_LDS_NUTS_PIECES work items per chain when there are more chains than chain slots and 8+ draws; never
# with bounds or a dense precond_mat).  The four waves of a chain tile store their quarters of theta, the tile's wave 0 publishes the chain at the next vote.
def _lds_problem(kind, d, n_rows, seed):
    bs = ((16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128) if kind == "logistic" else (48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128))
    if kind == "logistic":
        X, y = synth.logistic_problem(d, n_rows, seed=seed)
        return mcmc_amd.TARGET_LOGISTIC, dict(X=X, y=y), orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=bs, eta_chains=2), bs
    prec = synth.dense_gaussian_precision(d, seed=seed)
    return mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=prec), orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, blocks=4, block_size=bs), bs


@pytest.mark.parametrize("kind,d,n_rows,C,cap,burn,keep,adapt,depth,diag", [
    ("logistic", 100, 16, 150, 1, 10, 9, 10, 5, False),    # 19 draws: pieces of 5, 5, 5, 4; the window ends at a piece boundary; 4-5 chains per slot
    ("dense", 160, 0, 100, 1, 0, 17, 0, 5, False),         # no adaptation
    ("logistic", 72, 24, 200, 2, 20, 20, 33, 4, True),     # two workgroups exchange pieces; pieces of 10, the window ends inside the last piece but one; diagonal M
    ("logistic", 300, 20, 70, 1, 9, 8, 7, 4, False),       # the wide tiles (the momentum leaves the registers for the evaluation)
    ("dense", 512, 0, 70, 1, 6, 10, 16, 4, True),
    ("logistic", 100, 16, 150, 1, 4, 4, 4, 5, False),      # the shortest run that is cut: 8 draws, pieces of 2
])
def test_lds_nuts_runs_cut_into_pieces_match_the_oracle(kind, d, n_rows, C, cap, burn, keep, adapt, depth, diag, grid_cap):
    tk, tkw, spec, bs = _lds_problem(kind, d, n_rows, seed=d + 5)
    init = synth.initial_states(C, d, seed=d + 1) * (0.1 if kind == "logistic" else 0.5)
    init[40] = 1e200                                  # flagged at its first evaluation (piece 0): PQ_GONE in every later queue
    M = np.diag(np.random.default_rng(d).uniform(0.4, 2.5, d)) if diag else None
    kw = dict(precond_mat=M) if diag else {}
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=adapt, max_tree_depth=depth,
                                   step_size=1.0 if adapt else 0.05, **kw)
    grid_cap(cap)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=3, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    s = orc.make_settings(seed=7, n_burnin=burn, n_keep=keep, n_adapt=adapt, max_depth=depth, step=1.0 if adapt else 0.05, W=4, blocks=4, block_size=bs, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=3)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True)
    assert (g["n_exec"] <= g["n_leap"]).all()


@pytest.mark.parametrize("kind,d,n_rows,C,burn,keep,adapt,mass", [("logistic", 100, 16, 150, 10, 9, 10, False), ("dense", 192, 0, 100, 3, 6, 12, True), ("logistic", 300, 20, 70, 9, 8, 7, True)])
def test_lds_nuts_with_bounds_is_cut_into_pieces_too(kind, d, n_rows, C, burn, keep, adapt, mass, grid_cap):
    """settings.vals_bound: the hand-over carries theta in the TRANSFORMED space, as the chain holds it (through inv_transform and transform it would be rounded twice);
    mi_chains.theta holds constrained values again at the end; a chain flagged in piece 0"""
    tk, tkw, spec, bs = _lds_problem(kind, d, n_rows, seed=d + 2)
    rng = np.random.default_rng(d)
    bk = np.where(rng.random(d) < 0.4, rng.integers(2, 5, d), 1); bk[0] = 4; bk[d - 1] = 2
    lb = np.where((bk == 2) | (bk == 4), -1.5, -np.inf); ub = np.where((bk == 3) | (bk == 4), 2.0, np.inf)
    init = np.clip(synth.initial_states(C, d, seed=d + 4) * (0.1 if kind == "logistic" else 0.5), -1.0, 1.5)
    init[40, 1:d - 1] *= 1e200
    kw, okw = dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub), dict(lower=lb, upper=ub)
    if mass:
        M = np.diag(rng.uniform(0.4, 2.5, d)); kw["precond_mat"] = M; okw["precond"] = M
    st = mcmc_amd.default_settings(rng_seed_value=17, n_burnin_draws=burn, n_keep_draws=keep, step_size=0.05, n_adapt_draws=adapt, max_tree_depth=4, **kw)
    grid_cap(1)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=6, want_adapt_state=True, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<") and mcmc_amd.last_kernel().endswith("true, true>")
    s = orc.make_settings(seed=17, n_burnin=burn, n_keep=keep, step=0.05, n_adapt=adapt, max_depth=4, W=4, hoist=1, blocks=4, block_size=bs, **okw)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=6)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    assert np.array_equal(g["theta"], o_draws[-1], equal_nan=True)
    grid_cap(0)                                          # ... and what the call exports is what the uncut run exports
    u_draws, u = mcmc_amd.sample("nuts", tk, init, st, chain0=6, want_adapt_state=True, **tkw)
    assert np.array_equal(u_draws, g_draws, equal_nan=True)
    for k in ("adapt_state", "eps", "theta", "n_leap", "n_accept", "depth"):
        assert np.array_equal(u[k], g[k], equal_nan=True), k


@pytest.mark.parametrize("kind,d,n_rows,C,burn,keep,adapt", [("logistic", 100, 16, 100, 6, 6, 8), ("dense", 160, 0, 100, 10, 9, 10), ("dense", 512, 0, 70, 3, 6, 12)])
def test_lds_nuts_with_a_dense_precond_mat_is_cut_into_pieces_too(kind, d, n_rows, C, burn, keep, adapt, grid_cap):
    """logit_lds_kernel<., nuts, ., false, false, true> (logistic_nuts_dense_m.hip): the same cut; two chains leave the finite regime at once (PQ_GONE in every later queue, replayed with the same matrices)"""
    tk, tkw, spec, bs = _lds_problem(kind, d, n_rows, seed=d)
    rng = np.random.default_rng(d + 11)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    M = A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=d + 1) * (0.1 if kind == "logistic" else 0.5)
    init[5] *= 1e200; init[49, 3] = np.inf
    eps = 0.05 if kind == "logistic" else 0.1
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=adapt, max_tree_depth=4, step_size=eps, precond_mat=M)
    grid_cap(1)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, chain0=3, want_adapt_state=True, **tkw)
    kern = mcmc_amd.last_kernel()
    assert kern.startswith("logit_lds_kernel<") and kern.endswith("false, false, true>"), kern
    s = orc.make_settings(seed=7, n_burnin=burn, n_keep=keep, n_adapt=adapt, max_depth=4, step=eps, W=4, hoist=1, precond=M, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=3)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
    grid_cap(0)                                          # ... and what the call exports is what the uncut run exports
    u_draws, u = mcmc_amd.sample("nuts", tk, init, st, chain0=3, want_adapt_state=True, **tkw)
    assert np.array_equal(u_draws, g_draws, equal_nan=True)
    for k in ("adapt_state", "eps", "theta", "n_leap", "n_accept", "depth"):
        assert np.array_equal(u[k], g[k], equal_nan=True), k


def test_lds_nuts_a_chain_flagged_in_a_later_piece_is_replayed_from_its_initial_values(grid_cap):
    """as test_a_chain_flagged_in_a_later_piece_is_replayed_from_its_initial_values, on the dense Gaussian of nuts_lds.hpp"""
    d, C = 160, 100
    tk, tkw, spec, bs = _lds_problem("dense", d, 0, seed=3)
    tkw = dict(prec=tkw["prec"] * 1.0e-100)
    spec = orc.TargetSpec(orc.TARGET_DENSE, d, prec=tkw["prec"], W=4, blocks=4, block_size=bs)
    init = synth.initial_states(C, d, seed=19)
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=0, n_keep_draws=20, n_adapt_draws=20, max_tree_depth=5, step_size=0.1, gamma_val=2.3e-4)
    grid_cap(1)
    g_draws, g = mcmc_amd.sample("nuts", tk, init, st, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    s = orc.make_settings(seed=6, n_burnin=0, n_keep=20, step=0.1, n_adapt=20, max_depth=5, W=4, gamma=2.3e-4, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=0)
    assert np.isinf(o["eps"]).sum() > C // 2
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    assert np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True) and np.array_equal(g["n_accept"], o["n_accept"])


def test_lds_nuts_a_continuation_call_is_cut_into_pieces_too(grid_cap):
    d, C, n_adapt = 100, 120, 14
    tk, tkw, spec, bs = _lds_problem("logistic", d, 30, seed=8)
    init = synth.initial_states(C, d, seed=8) * 0.1
    S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=n_adapt, max_tree_depth=4)
    grid_cap(1)
    whole, w = mcmc_amd.sample("nuts", tk, init, S(0, 30), want_adapt_state=True, **tkw)
    a_draws, a = mcmc_amd.sample("nuts", tk, init, S(0, 6), want_adapt_state=True, **tkw)                  # (6 draws: in one piece)
    b_draws, b = mcmc_amd.sample("nuts", tk, a["theta"].T, S(0, 24), draw0=6, step_size_in=a["eps"], adapt_state_in=a["adapt_state"], want_adapt_state=True, **tkw)
    assert np.array_equal(np.concatenate([a_draws, b_draws], axis=0), whole)
    assert np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"]) and np.array_equal(b["eps"], w["eps"]) and np.array_equal(b["adapt_state"], w["adapt_state"])
    s = orc.make_settings(seed=99, n_burnin=0, n_keep=30, n_adapt=n_adapt, max_depth=4, step=1.0, W=4, blocks=4, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, spec, init, s, chain0=0)
    assert np.array_equal(whole, o_draws) and np.array_equal(w["eps"], o["eps"])


@pytest.mark.parametrize("kind,d,n_rows,C", [("memo", 64, 0, 150), ("logistic", 100, 16, 100), ("dense", 160, 0, 100)])
@pytest.mark.parametrize("adapt", [0, 7, 30])
def test_what_a_call_exports_does_not_depend_on_the_cut(kind, d, n_rows, C, adapt, grid_cap):
    """mi_chains.nuts_adapt_state (h, eps_bar, mu) is dead weight for the draws behind the adaptation window, but a piece that starts there must still carry it: the
    run cut into pieces (grid cap 1) exports what the uncut run (no cap: fewer chains than chain slots) exports -- found by tests/fuzz_nuts_lds.py with a grid cap"""
    if kind == "memo":
        tk, tkw = mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=synth.dense_gaussian_precision(d, seed=6))
    else:
        tk, tkw, _, _ = _lds_problem(kind, d, n_rows, seed=d)
    init = synth.initial_states(C, d, seed=5) * (0.1 if kind == "logistic" else 0.5)
    st = mcmc_amd.default_settings(rng_seed_value=31, n_burnin_draws=8, n_keep_draws=9, n_adapt_draws=adapt, max_tree_depth=5, step_size=0.3)
    grid_cap(0)
    u_draws, u = mcmc_amd.sample("nuts", tk, init, st, want_adapt_state=True, **tkw)
    name = mcmc_amd.last_kernel()
    assert name.startswith("nuts_gauss_memo_kernel" if kind == "memo" else "logit_lds_kernel<")
    grid_cap(1)
    c_draws, c = mcmc_amd.sample("nuts", tk, init, st, want_adapt_state=True, **tkw)
    assert mcmc_amd.last_kernel() == name
    assert np.array_equal(c_draws, u_draws)
    for k in ("adapt_state", "eps", "theta", "n_leap", "n_accept", "depth"):
        assert np.array_equal(c[k], u[k]), k
