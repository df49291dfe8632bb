"""Oracle hardening (parity is unpinned -- the reference ships no vectors and cannot be built here).  What CAN be removed is
"the C oracle and the kernels share one misreading": independent numpy transcriptions of mcmc::hmc and mcmc::mala, written from the
reference's sources as they read (ref: src/hmc.cpp:84-205, src/mala.cpp:84-186, include/mcmc/mala.ipp:30-70, include/stats/dmvnorm.hpp:28-54,
include/misc/{determine_bounds_type,transform_vals,log_jacobian,inv_jacobian_adjust}.hpp) with numpy's own linear algebra (inv, cholesky,
slogdet -- not the oracle's Gauss-Jordan), fed the SAME normal / uniform tapes.  They must make the same accept decisions as the C oracle
and agree on the draws to rounding (1e-9): control flow, formulas, call order and RNG consumption are then stated twice, independently.
Bounds of every type, diagonal and dense precond_mat.  (tests/test_oracle_structure.py does the same for nuts.)"""
import math

import numpy as np
import pytest

import orc
from mcmc_amd import synth

EPS_DBL = np.finfo(np.float64).eps                          # mcmc_options.hpp:103


def _bounds_type(lb, ub):                                   # determine_bounds_type.hpp:27-57
    return [4 if (math.isfinite(l) and math.isfinite(u)) else 2 if math.isfinite(l) else 3 if math.isfinite(u) else 1 for l, u in zip(lb, ub)]


def _transform(x, bt, lb, ub):                              # transform_vals.hpp:25-58
    out = np.array(x, dtype=np.float64)
    for i, t in enumerate(bt):
        if t == 2: out[i] = math.log(x[i] - lb[i] + EPS_DBL)
        elif t == 3: out[i] = -math.log(ub[i] - x[i] + EPS_DBL)
        elif t == 4: out[i] = math.log(x[i] - lb[i] + EPS_DBL) - math.log(ub[i] - x[i] + EPS_DBL)
    return out


def _inv_transform(v, bt, lb, ub):                          # transform_vals.hpp:60-118 (finite inputs)
    out = np.array(v, dtype=np.float64)
    for i, t in enumerate(bt):
        if t == 2: out[i] = lb[i] + EPS_DBL + math.exp(v[i])
        elif t == 3: out[i] = ub[i] - EPS_DBL - math.exp(-v[i])
        elif t == 4: out[i] = (lb[i] - EPS_DBL + (ub[i] + EPS_DBL) * math.exp(v[i])) / (1.0 + math.exp(v[i]))
    return out


def _log_jacobian(v, bt, lb, ub):                           # log_jacobian.hpp:25-57
    r = 0.0
    for i, t in enumerate(bt):
        if t == 2: r += v[i]
        elif t == 3: r += -v[i]
        elif t == 4: r += math.log(ub[i] - lb[i]) + v[i] - 2 * math.log(1 + math.exp(v[i]))
    return r


def _inv_jacobian(v, bt, lb, ub):                           # inv_jacobian_adjust.hpp:25-56
    J = np.eye(len(v))
    for i, t in enumerate(bt):
        if t == 2: J[i, i] = math.exp(-v[i])
        elif t == 3: J[i, i] = math.exp(v[i])
        elif t == 4:
            e = math.exp(v[i]); J[i, i] = ((e + 1) * (e + 1)) / (e * (ub[i] - lb[i]))
    return J


def _dmvnorm_log(x, mu, Sigma):                             # dmvnorm.hpp:28-54
    k = len(x)
    xc = x - mu
    quad = float(xc @ np.linalg.solve(Sigma, xc))
    return -0.5 * k * math.log(2 * math.pi) - 0.5 * (np.linalg.slogdet(Sigma)[1] + quad)


class _Target:
    def __init__(self, kind, d, seed):
        self.kind = kind
        if kind == "dense":
            self.P = synth.dense_gaussian_precision(d, seed=seed)
            self.spec = orc.TargetSpec(orc.TARGET_DENSE, d, prec=self.P, W=1)
        else:
            self.X, self.y = synth.logistic_problem(d, 23, seed=seed)
            self.spec = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=self.X, y=self.y, W=1)

    def __call__(self, x, want_grad):                       # the target_log_kernel callback: value and, when asked, gradient
        if self.kind == "dense":
            w = self.P @ x
            return -0.5 * float(x @ w), (-w if want_grad else None)
        eta = self.X @ x
        val = float(np.sum(self.y * eta - np.logaddexp(0.0, eta))) - 0.5 * float(x @ x)
        g = self.X.T @ (self.y - 1.0 / (1.0 + np.exp(-eta))) - x if want_grad else None
        return val, g


def _py_hmc(tgt, init, seed, n_burnin, n_keep, L, step, lb=None, ub=None, M=None):
    d = len(init)
    vb = lb is not None
    bt = _bounds_type(lb, ub) if vb else None
    Mm = np.eye(d) if M is None else M                      # hmc.cpp:57-59
    inv_M, sqrt_M = np.linalg.inv(Mm), np.linalg.cholesky(Mm)
    def box_log_kernel(v):                                  # :84-95
        return tgt(_inv_transform(v, bt, lb, ub), False)[0] + _log_jacobian(v, bt, lb, ub) if vb else tgt(v, False)[0]
    def mntm_update(pos, mntm):                             # :99-128
        if vb:
            g = tgt(_inv_transform(pos, bt, lb, ub), True)[1]
            return mntm + step * (_inv_jacobian(pos, bt, lb, ub) @ g) / 2
        return mntm + step * tgt(pos, True)[1] / 2
    first = _transform(init, bt, lb, ub) if vb else np.array(init, dtype=np.float64)     # :134-136
    prev_U = -box_log_kernel(first)
    prev_draw, rows, accepts = first.copy(), [], []
    for draw_ind in range(n_burnin + n_keep):
        new_mntm = sqrt_M @ orc.normal_vec(seed, 0, draw_ind, 0, d)                     # :156-158
        prev_K = float(new_mntm @ (inv_M @ new_mntm)) / 2
        new_draw = prev_draw.copy()
        for _ in range(L):                                  # :164-176
            new_mntm = mntm_update(new_draw, new_mntm)
            new_draw = new_draw + step * (inv_M @ new_mntm)
            new_mntm = mntm_update(new_draw, new_mntm)
        prop_U = -box_log_kernel(new_draw)
        if not math.isfinite(prop_U): prop_U = math.inf
        prop_K = float(new_mntm @ (inv_M @ new_mntm)) / 2
        comp_val = min(0.01, -(prop_U + prop_K) + (prev_U + prev_K))
        acc = orc.uniform(seed, 0, draw_ind, 0) < math.exp(comp_val)                     # :189-191
        if acc: prev_draw, prev_U = new_draw, prop_U
        accepts.append(int(acc))
        if draw_ind >= n_burnin: rows.append(prev_draw.copy())
    rows = np.array(rows)
    return (np.array([_inv_transform(r, bt, lb, ub) for r in rows]) if vb else rows), accepts      # :211-218


def _py_mala(tgt, init, seed, n_burnin, n_keep, step, lb=None, ub=None, M=None):
    d = len(init)
    vb = lb is not None
    bt = _bounds_type(lb, ub) if vb else None
    Mm = np.eye(d) if M is None else M                      # mala.cpp:57-58
    sqrt_M = np.linalg.cholesky(Mm)
    def box_log_kernel(v):
        return tgt(_inv_transform(v, bt, lb, ub), False)[0] + _log_jacobian(v, bt, lb, ub) if vb else tgt(v, False)[0]
    def mean_fn(v):                                         # :97-125; returns (mean, jacobian or None)
        if vb:
            g = tgt(_inv_transform(v, bt, lb, ub), True)[1]
            J = _inv_jacobian(v, bt, lb, ub)
            return v + step * step * (J @ (Mm @ g)) / 2, J
        return v + step * step * (Mm @ tgt(v, True)[1]) / 2, None
    def adjustment(prop, prev):                             # mala.ipp:30-70 (both densities with the PROPOSAL's Jacobian, as written)
        pm, pj = mean_fn(prop)
        vm, _ = mean_fn(prev)
        S = step * step * (pj @ Mm) if vb else step * step * Mm
        return _dmvnorm_log(prev, pm, S) - _dmvnorm_log(prop, vm, S)
    first = _transform(init, bt, lb, ub) if vb else np.array(init, dtype=np.float64)
    prev_LP = box_log_kernel(first)
    prev_draw, rows, accepts = first.copy(), [], []
    for draw_ind in range(n_burnin + n_keep):
        z = orc.normal_vec(seed, 0, draw_ind, 0, d)
        mean, J = mean_fn(prev_draw)
        new_draw = mean + step * (np.linalg.cholesky(J) @ (sqrt_M @ z)) if vb else mean + step * (sqrt_M @ z)   # :152-159
        prop_LP = box_log_kernel(new_draw)
        if not math.isfinite(prop_LP): prop_LP = -math.inf
        comp_val = min(0.01, prop_LP - prev_LP + adjustment(new_draw, prev_draw))       # :170
        acc = orc.uniform(seed, 0, draw_ind, 0) < math.exp(comp_val)
        if acc: prev_draw, prev_LP = new_draw, prop_LP
        accepts.append(int(acc))
        if draw_ind >= n_burnin: rows.append(prev_draw.copy())
    rows = np.array(rows)
    return (np.array([_inv_transform(r, bt, lb, ub) for r in rows]) if vb else rows), accepts


def _general(d, which, rng):
    lb = ub = M = None
    if "bounds" in which:
        kind = np.arange(d) % 4 + 1                         # all four bounds types
        lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    if "diag" in which: M = np.diag(rng.uniform(0.5, 2.0, d))
    if "dense" in which:
        A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.5, 2.0, d))
    return lb, ub, M


@pytest.mark.parametrize("kind", ["dense", "logit"])
@pytest.mark.parametrize("which", ["plain", "bounds", "diag", "dense", "bounds+dense"])
def test_python_transcription_of_hmc_agrees_with_the_oracle(kind, which):
    d, seed, n_burnin, n_keep, L, step = 6, 13, 3, 12, 4, 0.15
    rng = np.random.default_rng(5)
    tgt = _Target(kind, d, seed=4)
    lb, ub, M = _general(d, which, rng)
    init = np.clip(synth.initial_states(1, d, seed=2)[0] * 0.4, -1.0, 1.5)
    rows, accepts = _py_hmc(tgt, init, seed, n_burnin, n_keep, L, step, lb, ub, M)
    okw = {}
    if lb is not None: okw.update(lower=lb, upper=ub)
    if M is not None: okw.update(precond=M)
    st = orc.make_settings(seed=seed, n_burnin=n_burnin, n_keep=n_keep, n_leap=L, step=step, W=1, **okw)
    o_draws, o = orc.run_chain(orc.ALGO_HMC, tgt.spec, init, st, traces=True)
    assert list(o["accept"]) == accepts and 0 < sum(accepts)
    assert np.allclose(o_draws, rows, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("kind", ["dense", "logit"])
@pytest.mark.parametrize("which", ["plain", "bounds", "diag", "dense", "bounds+dense"])
def test_python_transcription_of_mala_agrees_with_the_oracle(kind, which):
    d, seed, n_burnin, n_keep, step = 5, 29, 3, 14, 0.12
    rng = np.random.default_rng(6)
    tgt = _Target(kind, d, seed=7)
    lb, ub, M = _general(d, which, rng)
    init = np.clip(synth.initial_states(1, d, seed=3)[0] * 0.4, -1.0, 1.5)
    rows, accepts = _py_mala(tgt, init, seed, n_burnin, n_keep, step, lb, ub, M)
    okw = {}
    if lb is not None: okw.update(lower=lb, upper=ub)
    if M is not None: okw.update(precond=M)
    st = orc.make_settings(seed=seed, n_burnin=n_burnin, n_keep=n_keep, step=step, W=1, **okw)
    o_draws, o = orc.run_chain(orc.ALGO_MALA, tgt.spec, init, st, traces=True)
    assert list(o["accept"]) == accepts and 0 < sum(accepts) < len(accepts) + 1
    assert np.allclose(o_draws, rows, rtol=1e-9, atol=1e-11)


# ---- mcmc::rmhmc (ref: src/rmhmc.cpp:84-290), with the metric tensors of the built-in targets written from their definitions
def _logit_tensor(tgt, x, want_deriv):
    """Fisher information of the logistic likelihood + the N(0, I) prior's precision: G = X' L X + I, L = diag(s (1 - s));
    dG/dbeta_i = X' diag(s (1 - s)(1 - 2 s) X[:, i]) X"""
    s = 1.0 / (1.0 + np.exp(-(tgt.X @ x)))
    lam = s * (1 - s)
    G = tgt.X.T @ (lam[:, None] * tgt.X) + np.eye(len(x))
    dG = [tgt.X.T @ ((lam * (1 - 2 * s) * tgt.X[:, i])[:, None] * tgt.X) for i in range(len(x))] if want_deriv else None
    return G, dG


def _py_rmhmc(tgt, tensor, init, seed, n_burnin, n_keep, L, step, n_fp, lb=None, ub=None):
    d = len(init)
    vb = lb is not None
    bt = _bounds_type(lb, ub) if vb else None
    nat = (lambda v: _inv_transform(v, bt, lb, ub)) if vb else (lambda v: v)
    def box_log_kernel(v):
        return tgt(nat(v), False)[0] + (_log_jacobian(v, bt, lb, ub) if vb else 0.0)
    def mntm_update(pos, mntm, inv_G, dG):                  # :99-150: step [J] grad_obj / 2
        g = tgt(nat(pos), True)[1]
        go = np.empty(d)
        for i in range(d):
            T = inv_G @ dG[i]
            go[i] = -g[i] + 0.5 * (np.trace(T) - float((T.T @ mntm) @ (inv_G @ mntm)))
        return step * ((_inv_jacobian(pos, bt, lb, ub) @ go) if vb else go) / 2
    box_tensor = lambda v, want: tensor(tgt, nat(v), want)  # :152-164
    first = _transform(init, bt, lb, ub) if vb else np.array(init, dtype=np.float64)
    new_tensor, new_deriv = box_tensor(first, True)
    prev_tensor, inv_new = new_tensor, np.linalg.inv(new_tensor)
    inv_prev, prev_deriv = inv_new, new_deriv
    cons = 0.5 * d * math.log(2 * math.pi)
    prev_U = cons - box_log_kernel(first) + 0.5 * np.linalg.slogdet(new_tensor)[1]       # :197
    prev_draw, rows, accepts = first.copy(), [], []
    for draw_ind in range(n_burnin + n_keep):
        new_mntm = np.linalg.cholesky(prev_tensor) @ orc.normal_vec(seed, 0, draw_ind, 0, d)     # :207-209
        prev_K = float(new_mntm @ (inv_prev @ new_mntm)) / 2
        new_draw = prev_draw.copy()
        for _ in range(L):
            prop_mntm = new_mntm.copy()
            for _k in range(n_fp):                          # :220-222
                prop_mntm = new_mntm + mntm_update(new_draw, prop_mntm, inv_prev, prev_deriv)
            new_mntm = prop_mntm
            prop_draw = new_draw.copy()
            for _k in range(n_fp):                          # :231-235
                inv_new = np.linalg.inv(box_tensor(prop_draw, False)[0])
                prop_draw = new_draw + 0.5 * step * ((inv_prev + inv_new) @ new_mntm)
            new_draw = prop_draw
            new_tensor, new_deriv = box_tensor(new_draw, True)                           # :239-240
            inv_new = np.linalg.inv(new_tensor)
            new_mntm = new_mntm + mntm_update(new_draw, new_mntm, inv_new, new_deriv)    # :244
        prop_U = cons - box_log_kernel(new_draw) + 0.5 * np.linalg.slogdet(new_tensor)[1]
        if not math.isfinite(prop_U): prop_U = math.inf
        prop_K = float(new_mntm @ (inv_new @ new_mntm)) / 2
        comp_val = min(0.01, -(prop_U + prop_K) + (prev_U + prev_K))
        acc = orc.uniform(seed, 0, draw_ind, 0) < math.exp(comp_val)
        if acc:
            prev_draw, prev_U = new_draw, prop_U
            prev_tensor, inv_prev, prev_deriv = new_tensor, inv_new, new_deriv
        accepts.append(int(acc))
        if draw_ind >= n_burnin: rows.append(prev_draw.copy())
    rows = np.array(rows)
    return (np.array([_inv_transform(r, bt, lb, ub) for r in rows]) if vb else rows), accepts


@pytest.mark.parametrize("which", ["plain", "bounds"])
@pytest.mark.parametrize("n_fp,L", [(1, 1), (3, 2), (5, 1)])
def test_python_transcription_of_rmhmc_agrees_with_the_oracle(which, n_fp, L):
    d, seed, n_burnin, n_keep, step = 4, 41, 2, 10, 0.1
    tgt = _Target("logit", d, seed=9)
    lb, ub, _ = _general(d, which, np.random.default_rng(1))
    init = np.clip(synth.initial_states(1, d, seed=5)[0] * 0.3, -1.0, 1.5)
    rows, accepts = _py_rmhmc(tgt, _logit_tensor, init, seed, n_burnin, n_keep, L, step, n_fp, lb, ub)
    okw = dict(lower=lb, upper=ub) if lb is not None else {}
    st = orc.make_settings(seed=seed, n_burnin=n_burnin, n_keep=n_keep, n_leap=L, step=step, n_fp=n_fp, W=1, **okw)
    o_draws, o = orc.run_chain(orc.ALGO_RMHMC, tgt.spec, init, st, traces=True)
    assert list(o["accept"]) == accepts and 0 < sum(accepts)
    assert np.allclose(o_draws, rows, rtol=1e-8, atol=1e-10)


# ---- mcmc::rwmh (ref: src/rwmh.cpp:84-151): new_draw = prev_draw + par_scale CHOL_LOWER(cov_mat) z, min(0, .) in the accept test
def _py_rwmh(tgt, init, seed, n_burnin, n_keep, par_scale, lb=None, ub=None, cov=None):
    d = len(init)
    vb = lb is not None
    bt = _bounds_type(lb, ub) if vb else None
    chol = par_scale * np.linalg.cholesky(np.eye(d) if cov is None else cov)
    def box_log_kernel(v):
        return tgt(_inv_transform(v, bt, lb, ub), False)[0] + _log_jacobian(v, bt, lb, ub) if vb else tgt(v, False)[0]
    first = _transform(init, bt, lb, ub) if vb else np.array(init, dtype=np.float64)
    prev_LP, prev_draw, rows, accepts = box_log_kernel(first), first.copy(), [], []
    for draw_ind in range(n_burnin + n_keep):
        new_draw = prev_draw + chol @ orc.normal_vec(seed, 0, draw_ind, 0, d)
        prop_LP = box_log_kernel(new_draw)
        if not math.isfinite(prop_LP): prop_LP = -math.inf
        acc = orc.uniform(seed, 0, draw_ind, 0) < math.exp(min(0.0, prop_LP - prev_LP))
        if acc: prev_draw, prev_LP = new_draw, prop_LP
        accepts.append(int(acc))
        if draw_ind >= n_burnin: rows.append(prev_draw.copy())
    rows = np.array(rows)
    return (np.array([_inv_transform(r, bt, lb, ub) for r in rows]) if vb else rows), accepts


@pytest.mark.parametrize("kind", ["dense", "logit"])
@pytest.mark.parametrize("which", ["plain", "bounds", "dense", "bounds+dense"])
def test_python_transcription_of_rwmh_agrees_with_the_oracle(kind, which):
    d, seed, n_burnin, n_keep, scale = 6, 53, 3, 25, 0.25
    tgt = _Target(kind, d, seed=4)
    lb, ub, cov = _general(d, which, np.random.default_rng(8))
    init = np.clip(synth.initial_states(1, d, seed=2)[0] * 0.4, -1.0, 1.5)
    rows, accepts = _py_rwmh(tgt, init, seed, n_burnin, n_keep, scale, lb, ub, cov)
    okw = {}
    if lb is not None: okw.update(lower=lb, upper=ub)
    if cov is not None: okw.update(precond=cov)
    st = orc.make_settings(seed=seed, n_burnin=n_burnin, n_keep=n_keep, step=scale, W=1, **okw)
    o_draws, o = orc.run_chain(orc.ALGO_RWMH, tgt.spec, init, st, traces=True)
    assert list(o["accept"]) == accepts and 0 < sum(accepts) < len(accepts)
    assert np.allclose(o_draws, rows, rtol=1e-10, atol=1e-12)


# ---- the non-finite regime.  The reference multiplies by its identity / diagonal matrices as DENSE products, where one
# non-finite entry makes 0 * inf = NaN in every other row.  numpy's matmul forms every product as well, so the transcriptions above show the
# same poisoning without having been told about it: NaN patterns and accept sequences must match the oracle's.
@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("which", ["plain", "diag"])
@pytest.mark.parametrize("bad", ["inf", "huge", "nan"])
def test_transcriptions_poison_like_the_oracle_in_the_non_finite_regime(algo, which, bad):
    d, seed = 5, 7
    tgt = _Target("dense", d, seed=3)
    _, _, M = _general(d, which, np.random.default_rng(2))
    init = synth.initial_states(1, d, seed=4)[0]
    init[2] = {"inf": np.inf, "huge": 1e300, "nan": np.nan}[bad]
    okw = dict(precond=M) if M is not None else {}
    with np.errstate(all="ignore"):
        if algo == "hmc":
            rows, accepts = _py_hmc(tgt, init, seed, 1, 6, 3, 0.2, M=M)
            st = orc.make_settings(seed=seed, n_burnin=1, n_keep=6, n_leap=3, step=0.2, W=1, **okw)
            o_draws, o = orc.run_chain(orc.ALGO_HMC, tgt.spec, init, st, traces=True)
        else:
            rows, accepts = _py_mala(tgt, init, seed, 1, 6, 0.2, M=M)
            st = orc.make_settings(seed=seed, n_burnin=1, n_keep=6, step=0.2, W=1, **okw)
            o_draws, o = orc.run_chain(orc.ALGO_MALA, tgt.spec, init, st, traces=True)
    assert list(o["accept"]) == accepts
    assert np.array_equal(np.isnan(o_draws), np.isnan(rows)) and np.array_equal(np.isinf(o_draws), np.isinf(rows))
    assert bad == "huge" or np.isnan(o_draws).any()      # inf / nan starts do reach the regime (1e300 stays finite here: an overflow-free giant)
    fin = np.isfinite(o_draws)
    assert np.allclose(o_draws[fin], rows[fin], rtol=1e-9)
