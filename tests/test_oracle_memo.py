"""CPU: the memoised evaluation of a NUTS doubling (oracle/mcmc_oracle.c: nuts_doubling_memo -- what mcmc_amd/csrc/nuts_memo.hpp runs on the
device) against the literal recursion (nuts_build_tree, ref: include/mcmc/nuts.ipp:97-241): identical draws, accepts, tree depths, reference
leapfrog counts, step sizes -- bit for bit -- while making far fewer leap_frog calls.  Every U-turn test the walk reads must have been
evaluated when its second trajectory point appeared (the oracle aborts otherwise)."""
import numpy as np
import pytest

import orc
from mcmc_amd import synth


def _both(kind, d, init, chain, **kw):
    prec = None
    if kind == orc.TARGET_DENSE:
        prec = synth.dense_gaussian_precision(d, seed=6)
    elif kind == orc.TARGET_DIAG:
        prec = synth.ill_conditioned_diag(d, 50.0)
    out = []
    for algo in (orc.ALGO_NUTS, orc.ALGO_NUTS_MEMO, orc.ALGO_NUTS_MEMO_XD):
        t = orc.TargetSpec(kind, d, prec=prec, W=4)
        s = orc.make_settings(W=4, chain_id=chain, **kw)
        out.append(orc.run_chain(algo, t, init, s, traces=True))
    return out


CASES = [
    # kind, d, burn, keep, adapt, max_depth, step
    (orc.TARGET_DENSE, 8, 5, 20, 15, 10, 1.0),
    (orc.TARGET_ISO, 3, 30, 30, 30, 10, 1.0),
    (orc.TARGET_DENSE, 33, 4, 8, 6, 4, 1.0),
    (orc.TARGET_DIAG, 20, 0, 12, 0, 6, 0.05),         # fixed small step: trees to the cap
    (orc.TARGET_DENSE, 16, 2, 10, 12, 1, 1.0),
    (orc.TARGET_DENSE, 24, 0, 4, 0, 10, 0.004),       # depth-10 trees: 1023 leaves on 175 points
    (orc.TARGET_DENSE, 12, 2, 3, 0, 0, 1.0),          # max_tree_depth = 0
    (orc.TARGET_DENSE, 40, 30, 10, 30, 10, 1.0),      # BASELINE configs[3] in small: adaptation from eps = 1, deep trees early on
]


@pytest.mark.parametrize("kind,d,burn,keep,adapt,depth,step", CASES)
def test_memoised_doubling_equals_the_recursion(kind, d, burn, keep, adapt, depth, step):
    """... and (round 6) memoised ACROSS the doublings of a draw as well (orc_nuts_memo_xd: a doubling re-uses the points the last doubling of its direction
    left, while no proposal was accepted in between): the same bits again, fewer leap_frog calls still"""
    tot_ref = tot_exec = tot_xd = 0
    for chain in range(6):
        init = synth.initial_states(1, d, seed=100 + chain)[0]
        (a, ia), (b, ib), (x, ix) = _both(kind, d, init, chain, seed=77, n_burnin=burn, n_keep=keep, n_adapt=adapt, max_depth=depth, step=step)
        for o, io in ((b, ib), (x, ix)):
            assert np.array_equal(a, o)
            for k in ("n_accept", "n_leap", "eps"):
                assert ia[k] == io[k], k
            for k in ("accept", "depth", "leaps", "eps_trace"):
                assert np.array_equal(ia[k], io[k]), k
        assert ix["n_exec"] <= ib["n_exec"] <= ib["n_leap"]
        tot_ref += ib["n_leap"]; tot_exec += ib["n_exec"]; tot_xd += ix["n_exec"]
    if depth >= 6 and kind != orc.TARGET_ISO:
        assert tot_exec < tot_ref                       # points are re-visited: fewer leap_frog calls than leaves
        assert tot_xd < tot_exec                        # ... and doublings of one direction share their trajectory


def test_memoised_doubling_in_the_non_finite_regime_and_with_bounds():
    d = 10
    prec = synth.dense_gaussian_precision(d, seed=6)
    lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 2 == 0, 2.0, np.inf)
    M = np.diag(np.linspace(0.5, 2.0, d))
    for chain, scale, kw in [(0, 1e200, {}), (1, 1.0, dict(lower=lb, upper=ub)), (2, 0.3, dict(precond=M)), (3, 1e160, dict(precond=M)),
                             (4, 0.5, dict(lower=lb, upper=ub, precond=M))]:
        init = np.clip(synth.initial_states(1, d, seed=chain)[0] * 0.3, -1.0, 1.5) * scale
        res = []
        for algo in (orc.ALGO_NUTS, orc.ALGO_NUTS_MEMO, orc.ALGO_NUTS_MEMO_XD):
            t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
            s = orc.make_settings(seed=5, n_burnin=6, n_keep=8, n_adapt=6, max_depth=7, step=0.3, W=4, chain_id=chain, **kw)
            res.append(orc.run_chain(algo, t, init, s, traces=True))
        (a, ia) = res[0]
        for b, ib in res[1:]:
            assert np.array_equal(a, b, equal_nan=True)
            assert ia["n_leap"] == ib["n_leap"] and np.array_equal(ia["depth"], ib["depth"]) and np.array_equal(ia["accept"], ib["accept"])
            assert (ia["eps"] == ib["eps"]) or (np.isnan(ia["eps"]) and np.isnan(ib["eps"]))


def test_the_points_of_a_doubling_follow_the_closed_form():
    """n(i) = 1 + sum over set bits k of i of (k + 1) from the start-leaf rule of nuts_dense.hpp; a level-l node's test spans l points"""
    def ctz(i): return (i & -i).bit_length() - 1
    n = [1]
    for i in range(1, 1024):
        c = ctz(i)
        n.append(n[i - 1 if c <= 1 else i - (1 << (c - 1))] + 1)
    for i in range(1024):
        assert n[i] == 1 + sum(k + 1 for k in range(10) if (i >> k) & 1)
        if i:
            c = ctz(i)
            assert n[i] - n[i - (1 << c)] == c + 1
    assert max(n) == 56 and len(set(n)) == 56


def test_the_device_formulas_of_the_memoised_tick():
    """include/mi_mcmc_engine/nuts_memo_core.hpp computes n(i) with five popcounts (the bit positions k grouped by the bits of k) and decides which
    (level, first point) pairs a doubling uses from the subset sums of consecutive integers; both restated here and checked against the definition."""
    pc = lambda x: bin(x).count("1")
    for i in range(1024):
        n_dev = 1 + pc(i) + pc(i & 0x2AA) + 2 * pc(i & 0xCC) + 4 * pc(i & 0xF0) + 8 * pc(i & 0x300)
        assert n_dev == 1 + sum(k + 1 for k in range(10) if (i >> k) & 1)

    def used(l, n1, j):                                    # memo_pair_used
        m = n1 - 1
        return any(t * (l + 1) + t * (t - 1) // 2 <= m <= t * j - t * (t - 1) // 2 for t in range(0, j - l + 1))

    npt = lambda i: 1 + sum(k + 1 for k in range(10) if (i >> k) & 1)
    for j in range(10):
        truth = {(l, npt(b)) for l in range(1, j + 1) for b in range(0, 1 << j, 1 << l)}      # first leaves of the level-l nodes
        for l in range(1, j + 1):
            for n1 in range(1, 48):
                assert used(l, n1, j) == ((l, n1) in truth), (j, l, n1)
        assert max(npt(i) for i in range(1 << j)) == 1 + j * (j + 1) // 2 <= 46
        # the test of a level-l node at first point n1 closes when point n1 + l appears: it exists by then (n1 + l <= the doubling's points)
        assert all(n1 + l <= 1 + j * (j + 1) // 2 for (l, n1) in truth)


def test_many_chains_through_run_many():
    d, C = 16, 24
    prec = synth.dense_gaussian_precision(d, seed=2)
    init = synth.initial_states(C, d, seed=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=9, n_burnin=20, n_keep=10, n_adapt=20, max_depth=10, W=4)
    a, ia = orc.run_many(orc.ALGO_NUTS, t, init, s, chain0=11)
    b, ib = orc.run_many(orc.ALGO_NUTS_MEMO, t, init, s, chain0=11)
    assert np.array_equal(a, b) and np.array_equal(ia["n_leap"], ib["n_leap"]) and np.array_equal(ia["eps"], ib["eps"])
