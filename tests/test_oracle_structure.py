"""Hardening of the parity-unpinned oracle (VERDICT r2 "Next" 8): checks that do not share the oracle's code.

(i)   NUTS structural KAT.  From the argument plumbing of ref: include/mcmc/nuts.ipp:159-240 (first child starts at the subtree's
      start, second child at the first child's FAR edge) the leaf start offsets of a depth-k subtree from offset o are
      starts(k, o) = starts(k-1, o) ++ starts(k-1, o + k), so a subtree touches the 1 + k(k+1)/2 contiguous offsets o+1 .. o+M_k
      (SURVEY App. B).  Observed on the oracle through a free-particle callback target (constant log kernel: the position moves
      by one step per leapfrog, no U-turn ever triggers, every energy test passes), whose call log IS the sequence of offsets.
(ii)  The 2-D hand-computed value of mala_prop_adjustment (ref: include/mcmc/mala.ipp:30-70).
(iii) An independently written Python transcription of nuts_build_tree (recursive, by reference, exactly as nuts.ipp:97-241 is
      written) and of the loop of src/nuts.cpp:199-310, fed the same normal / uniform tapes, against the C oracle: same tree
      depths, same leapfrog counts, same accept decisions, draws to 1e-10 (numpy sums are not fma chains, libm is not orc_math).
None of this pins parity to the reference (which cannot be built here); it removes "both sides share one misreading"."""
import ctypes as C
import math

import numpy as np
import pytest

import orc
from mcmc_amd import synth


# ------------------------------------------------------------------ (i) structural KAT
def _starts(k, o):
    return [o] if k == 0 else _starts(k - 1, o) + _starts(k - 1, o + k)


@pytest.mark.parametrize("max_depth", [1, 3, 6, 9])
def test_nuts_leaf_offsets_follow_the_crossed_edge_plumbing(max_depth):
    log = []                                    # (had_grad, position) per callback
    lam = 1.0e-10                               # a 1-D Gaussian so flat that 2^9 steps of 0.25 stay far from a U-turn (omega t = 0.0013)

    @C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
    def flat_gaussian(vals, grad_out, data):
        log.append((bool(grad_out), vals[0]))
        if grad_out:
            grad_out[0] = -lam * vals[0]
        return -0.5 * lam * vals[0] * vals[0]

    tgt = orc.TargetSpec(orc.TARGET_ISO, 1, W=1)
    # n_adapt = 0: draw 0 runs at the searched step size (huge on this target), every later draw at epsilon_bar = 0.25
    st = orc.make_settings(seed=5, n_burnin=0, n_keep=4, n_adapt=0, max_depth=max_depth, step=0.25, W=1)
    draws, info = orc.run_chain(orc.ALGO_NUTS, tgt, np.array([0.125]), st, traces=True, kernel=flat_gaussian)
    # (a doubling's subtree never fails here -- the trajectory is straight -- but the TOP-level test (src/nuts.cpp:286-289) can: every
    #  doubling restarts from the current prev_draw (:241-256) while the far side keeps the edge of an earlier origin, so it may end a
    #  draw before max_tree_depth; the depth trace says how many doublings each draw made)
    depth = [int(x) for x in info["depth"]]
    assert max(depth[1:]) == max_depth
    # parsed from the end (the search and draw 0 come first and are never reached): per draw, per doubling j: 2^j leaves of
    # (grad at start, grad at end, value at end) [+ one value call when the top level accepts the proposal, src/nuts.cpp:264]
    calls = log
    i = len(calls)
    doublings = []
    for draw in reversed(range(1, 4)):
        for j in reversed(range(depth[draw])):
            if not calls[i - 1][0] and not calls[i - 2][0]:                 # ... a leaf's value call, then the accept's
                i -= 1
            leaves = []
            for _ in range(2 ** j):
                (g0, x0), (g1, x1), (g2, x2) = calls[i - 3], calls[i - 2], calls[i - 1]
                assert g0 and g1 and not g2 and x1 == x2, "a leaf is grad(start), grad(end), value(end)"
                leaves.append((x0, x1))
                i -= 3
            doublings.append((j, leaves[::-1]))
    for j, leaves in doublings:
        # offsets from the log alone: the first leaf starts at offset 0; a leaf ends one step after its start; every later start is
        # a state seen before (revisited offsets are recomputed bit-identically)
        offset_of = {leaves[0][0]: 0}
        off_start, off_end = [], []
        for a, b_ in leaves:
            assert a in offset_of, "a leaf starts from the doubling's origin or from a state some earlier leaf produced"
            o = offset_of[a]
            assert offset_of.setdefault(b_, o + 1) == o + 1, "the state at an offset does not depend on the path to it"
            off_start.append(o); off_end.append(o + 1)
        assert off_start == _starts(j, 0), f"depth {j}: leaf start offsets"
        M = 1 + j * (j + 1) // 2
        assert sorted(set(off_end)) == list(range(1, M + 1)), f"depth {j}: a subtree touches offsets 1..{M}"
        assert off_end[-1] == M                              # the last leaf ends at the far edge ... which is offset j + 1 only for j <= 1


def test_nuts_far_and_near_edges_sit_at_offsets_k_plus_1_and_1():
    """edges of a depth-k subtree from offset o: near = state(o + 1), far = state(o + k + 1) -- read off the second children's start
    offsets: the second child of a depth-k node starts at its first child's far edge, o + (k - 1) + 1 = o + k"""
    for k in range(1, 10):
        s = _starts(k, 0)
        assert s[2 ** (k - 1)] == k                          # second child of the root starts at offset k
        assert s[0] == 0 and s[1] == (1 if k >= 1 else 0)


# ------------------------------------------------------------------ (ii) mala_prop_adjustment by hand
def test_mala_prop_adjustment_two_dimensional_hand_value():
    """iso Gaussian, M = I: mu(v) = v + eps^2 grad(v) / 2 = v (1 - eps^2 / 2); Sigma = eps^2 I, so the log-determinant cancels and
    adj = [ |prop - mu(prev)|^2 - |prev - mu(prop)|^2 ] / (2 eps^2).  Checked through the accept decision's inputs: the oracle's
    exported evaluation of the adjustment."""
    eps = 0.5
    prev = np.array([1.0, -2.0]); prop = np.array([0.5, 0.25])
    c = 1.0 - eps * eps / 2.0                                # 0.875
    mu_prev, mu_prop = prev * c, prop * c                    # (0.875, -1.75), (0.4375, 0.21875)
    q_b = ((prop - mu_prev) ** 2).sum()                      # 0.140625 + 4.0 = 4.140625
    q_a = ((prev - mu_prop) ** 2).sum()                      # 0.31640625 + 4.9228515625 = 5.2392578125
    assert q_b == 4.140625 and q_a == 5.2392578125           # exact in binary
    by_hand = (q_b - q_a) / (2 * eps * eps)                  # -2.197265625
    assert by_hand == -2.197265625
    tgt = orc.TargetSpec(orc.TARGET_ISO, 2, W=1)
    for hoist in (0, 1):
        s = orc.make_settings(step=eps, W=1, hoist=hoist)
        got = orc.mala_prop_adjustment(tgt, s, prop, prev)
        assert abs(got - by_hand) < 1e-13, (got, by_hand)


# ------------------------------------------------------------------ (iii) independent transcription of NUTS on tapes
class _PyNuts:
    """nuts.ipp / nuts.cpp as written, numpy arrays standing in for ColVec_t, passed and assigned by reference"""

    def __init__(self, P, seed, chain, max_depth, eps0, n_adapt, delta=0.55, gamma=0.05, t0=10.0, kappa=0.75):
        self.P, self.seed, self.chain = P, seed, chain
        self.max_depth, self.eps0, self.n_adapt = max_depth, eps0, n_adapt
        self.delta, self.gamma, self.t0, self.kappa = delta, gamma, t0, kappa
        self.n_leap = 0

    def kernel(self, x, want_grad):
        w = self.P @ x
        return -0.5 * float(x @ w), (-w if want_grad else None)

    def leap_frog(self, step, draw, mntm):                  # one step, in place (nuts.cpp:139-154 with hmc.cpp:124-126)
        _, g = self.kernel(draw, True)
        mntm += step * g / 2
        draw += step * mntm
        _, g = self.kernel(draw, True)
        mntm += step * g / 2
        self.n_leap += 1

    def runif(self):
        u = orc.uniform(self.seed, self.chain, self.draw_ind, self.uslot)
        self.uslot += 1
        return u

    def build_tree(self, v, eps, log_u, prev_U, prev_K, draw_vec, mntm_vec, depth, new_draw, pos, neg, mpos, mneg):
        if depth == 0:
            new_draw[:] = draw_vec
            new_mntm = mntm_vec.copy()
            self.leap_frog(v * eps, new_draw, new_mntm)
            prop_U = -self.kernel(new_draw, False)[0]
            if not math.isfinite(prop_U):
                prop_U = math.inf
            prop_K = float(new_mntm @ new_mntm) / 2
            n = int(log_u <= -prop_U - prop_K)
            s = int(log_u < 1000 - prop_U - prop_K)
            pos[:] = new_draw; neg[:] = new_draw; mpos[:] = new_mntm; mneg[:] = new_mntm
            return n, s, math.exp(min(0.0, -(prop_U + prop_K) + (prev_U + prev_K))), 1
        new_draw_p = np.empty_like(draw_vec)
        n_p, s_p, a_p, na_p = self.build_tree(v, eps, log_u, prev_U, prev_K, draw_vec, mntm_vec, depth - 1, new_draw_p, pos, neg, mpos, mneg)
        if s_p == 1:
            new_draw_pp = np.empty_like(draw_vec)
            if v == -1:
                dummy_draw, dummy_mntm, draw_neg, mntm_neg = pos.copy(), mpos.copy(), neg.copy(), mneg.copy()
                n_pp, s_pp, a_pp, na_pp = self.build_tree(v, eps, log_u, prev_U, prev_K, draw_neg, mntm_neg, depth - 1,
                                                          new_draw_pp, neg, dummy_draw, mneg, dummy_mntm)
            else:
                dummy_draw, dummy_mntm, draw_pos, mntm_pos = neg.copy(), mneg.copy(), pos.copy(), mpos.copy()
                n_pp, s_pp, a_pp, na_pp = self.build_tree(v, eps, log_u, prev_U, prev_K, draw_pos, mntm_pos, depth - 1,
                                                          new_draw_pp, dummy_draw, pos, dummy_mntm, mpos)
            with np.errstate(divide="ignore", invalid="ignore"):
                prob = np.float64(n_pp) / np.float64(n_p + n_pp)
            z = self.runif()
            if z < prob:
                new_draw_p = new_draw_pp
            n_p += n_pp; a_p += a_pp; na_p += na_pp
            c1 = int(float((pos - neg) @ mneg) >= 0)
            c2 = int(float((pos - neg) @ mpos) >= 0)
            s_p = s_pp * c1 * c2
        new_draw[:] = new_draw_p
        return n_p, s_p, a_p, na_p

    def find_initial_step_size(self, draw_vec, mntm_vec):
        step = 1.0
        prev_U = -self.kernel(draw_vec, False)[0]
        prev_K = float(mntm_vec @ mntm_vec) / 2
        nd, nm = draw_vec.copy(), mntm_vec.copy()
        self.leap_frog(step, nd, nm)
        H = lambda: -(-self.kernel(nd, False)[0] + float(nm @ nm) / 2) + (prev_U + prev_K)
        a = 2 * int(H() > math.log(0.5)) - 1
        cond = H() > -math.log(2)
        while cond:
            step *= 2.0 ** a
            self.leap_frog(step, nd, nm)
            a = 2 * int(H() > math.log(0.5)) - 1
            cond = H() > -math.log(2)
        return step

    def run(self, initial, n_burnin, n_keep):
        d = len(initial)
        first = np.array(initial, dtype=np.float64)
        mntm_vec = orc.normal_vec(self.seed, self.chain, 0, 2, d)           # STREAM_INIT
        step = self.find_initial_step_size(first, mntm_vec)
        mu = math.log(10 * step)
        h = 0.0
        eps_bar = self.eps0
        prev_U = -self.kernel(first, False)[0]
        prev_draw = first.copy()
        draw_pos, draw_neg, mntm_pos, mntm_neg = first.copy(), first.copy(), mntm_vec.copy(), mntm_vec.copy()
        new_draw = first.copy()
        rows, depths, leaps, accepts = [], [], [], []
        for draw_ind in range(n_burnin + n_keep):
            self.draw_ind, self.uslot = draw_ind, 0
            leap0 = self.n_leap
            mntm_vec = orc.normal_vec(self.seed, self.chain, draw_ind, 0, d)  # STREAM_NORMAL
            prev_K = float(mntm_vec @ mntm_vec) / 2
            log_u = math.log(self.runif()) - prev_U - prev_K
            new_draw[:] = prev_draw; draw_pos[:] = prev_draw; draw_neg[:] = prev_draw; mntm_pos[:] = mntm_vec; mntm_neg[:] = mntm_vec
            depth, n_val, s_val, alpha, n_alpha, good = 0, 1, 1, 0.0, 0, 0
            while s_val == 1 and depth < self.max_depth:
                v = -1 if self.runif() <= 0.5 else 1
                start = prev_draw          # passed by const reference: the callee does not modify it
                if v == -1:
                    dd, dm = draw_pos.copy(), mntm_pos.copy()
                    n_p, s_p, alpha, n_alpha = self.build_tree(v, step, log_u, prev_U, prev_K, start, mntm_vec, depth, new_draw, dd, draw_neg, dm, mntm_neg)
                else:
                    dd, dm = draw_neg.copy(), mntm_neg.copy()
                    n_p, s_p, alpha, n_alpha = self.build_tree(v, step, log_u, prev_U, prev_K, start, mntm_vec, depth, new_draw, draw_pos, dd, mntm_pos, dm)
                if s_p == 1:
                    if self.runif() < np.float64(n_p) / np.float64(n_val):
                        prop_U = -self.kernel(new_draw, False)[0]
                        if not math.isfinite(prop_U):
                            prop_U = math.inf
                        prev_draw = new_draw.copy()
                        prev_U = prop_U
                        good = 1
                n_val += n_p
                depth += 1
                c1 = int(float((draw_pos - draw_neg) @ mntm_neg) >= 0)
                c2 = int(float((draw_pos - draw_neg) @ mntm_pos) >= 0)
                s_val = s_p * c1 * c2
            if draw_ind < self.n_adapt:
                h += (1 / (draw_ind + 1 + self.t0)) * (self.delta - (alpha / n_alpha) - h)
                step = math.exp(mu - h * math.sqrt(draw_ind + 1) / self.gamma)
                eps_bar *= math.exp((draw_ind + 1) ** (-self.kappa) * (math.log(step) - math.log(eps_bar)))
            else:
                step = eps_bar
            depths.append(depth); leaps.append(self.n_leap - leap0); accepts.append(good)
            if draw_ind >= n_burnin:
                rows.append(prev_draw.copy())
        return np.array(rows), depths, leaps, accepts, step


@pytest.mark.parametrize("d,max_depth,n_adapt,seed", [(3, 5, 0, 11), (5, 6, 4, 3), (8, 4, 10, 21), (2, 7, 0, 8)])
def test_python_transcription_of_nuts_agrees_with_the_oracle(d, max_depth, n_adapt, seed):
    P = synth.dense_gaussian_precision(d, seed=seed)
    init = synth.initial_states(1, d, seed=seed + 1)[0]
    n_burnin, n_keep = 4, 6
    py = _PyNuts(P, seed=seed, chain=0, max_depth=max_depth, eps0=0.3, n_adapt=n_adapt)
    rows, depths, leaps, accepts, step = py.run(init, n_burnin, n_keep)
    tgt = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P, W=1)
    st = orc.make_settings(seed=seed, n_burnin=n_burnin, n_keep=n_keep, n_adapt=n_adapt, max_depth=max_depth, step=0.3, W=1)
    o_draws, o = orc.run_chain(orc.ALGO_NUTS, tgt, init, st, traces=True)
    assert list(o["depth"]) == depths, "tree depth per draw"
    assert list(o["leaps"]) == leaps, "leapfrog steps per draw"
    assert list(o["accept"]) == accepts, "top-level accepts"
    assert int(o["n_leap"]) == py.n_leap, "leapfrog steps including the step-size search"
    assert np.allclose(o_draws, rows, rtol=1e-10, atol=1e-12)
    assert abs(o["eps"] - step) <= 1e-10 * abs(step)
