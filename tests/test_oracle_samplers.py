"""CPU: the oracle's samplers -- structural facts, closed forms and statistics (not gpu).

The reference ships no golden vectors; what pins the restatement here:
  * callback counts per draw that were OBSERVED on the reference's own src/{hmc,mala}.cpp during the
    survey: HMC 2*L gradient + 1 value per draw (+1 value at setup), MALA 3 gradient
    + 1 value per draw (+1 at setup); NUTS one gradient pair + one value per executed leapfrog;
  * closed forms (leapfrog on a Gaussian is linear; dmvnorm against scipy);
  * the plumbing/statistics contract of examples/eigen/hmc_normal.cpp:83-118.
"""
from fractions import Fraction

import numpy as np
import pytest

import orc
from mcmc_amd import synth


def test_hmc_callback_counts_match_reference_observation():
    t = orc.TargetSpec(orc.TARGET_ISO, 3)
    s = orc.make_settings(seed=1234, n_burnin=1000, n_keep=1000, n_leap=10, step=0.2)
    dr, info = orc.run_chain(orc.ALGO_HMC, t, np.ones(3), s)
    assert t.c.n_grad_calls == 2 * 10 * 2000          # 40 000
    assert t.c.n_value_calls == 2000 + 1              # 2 001
    assert info["n_leap"] == 10 * 2000
    assert dr.shape == (1000, 3)
    assert 0.95 < info["n_accept"] / 1000 <= 1.0      # acc 0.996 on the reference
    assert np.abs(dr.mean(0)).max() < 0.15 and np.abs(dr.var(0) - 1).max() < 0.2


def test_mala_callback_counts_match_reference_observation():
    t = orc.TargetSpec(orc.TARGET_ISO, 3)
    s = orc.make_settings(seed=1234, n_burnin=1000, n_keep=1000, step=0.5, hoist=0)
    dr, info = orc.run_chain(orc.ALGO_MALA, t, np.ones(3), s)
    assert t.c.n_grad_calls == 3 * 2000               # 6 000 (3 per draw)
    assert t.c.n_value_calls == 2000 + 1
    assert 0.5 < info["n_accept"] / 1000 <= 1.0


def test_mala_hoisting_the_factorisation_does_not_change_bits():
    P = synth.dense_gaussian_precision(6, seed=8)
    M = np.linalg.inv(P)
    M = 0.5 * (M + M.T)
    init = np.arange(6) * 0.1
    out = []
    for hoist in (0, 1):
        t = orc.TargetSpec(orc.TARGET_DENSE, 6, prec=P)
        s = orc.make_settings(seed=5, n_burnin=5, n_keep=40, step=0.3, precond=M, hoist=hoist)
        out.append(orc.run_chain(orc.ALGO_MALA, t, init, s, traces=True))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1]["accept"], out[1][1]["accept"])


def test_nuts_work_accounting_and_traces():
    t = orc.TargetSpec(orc.TARGET_ISO, 3)
    s = orc.make_settings(seed=1234, n_burnin=1000, n_keep=1000, step=1.0)
    dr, info = orc.run_chain(orc.ALGO_NUTS, t, np.ones(3), s, traces=True)
    # every executed leapfrog = 2 gradient callbacks; each tree leaf adds one value callback
    assert t.c.n_grad_calls == 2 * info["n_leap"]
    # leapfrogs per draw = sum over doublings of 2^j when no subtree stops early, fewer otherwise
    assert np.all(info["leaps"] <= 2 ** info["depth"].astype(np.int64) - 1)
    assert np.all(info["leaps"] >= info["depth"])
    assert info["depth"].max() <= 10
    # SURVEY 8(c) observed 2.5 leapfrogs/draw and acc ~0.70 on the reference for this target
    assert 1.5 < info["n_leap"] / 2000 < 4.0
    assert 0.5 < info["n_accept"] / 1000 < 0.95
    assert np.abs(dr.mean(0)).max() < 0.2 and np.abs(dr.var(0) - 1).max() < 0.3
    # dual averaging moved the step size and froze it after n_adapt (=1000 = all burn-in) draws
    assert len(set(info["eps_trace"][1001:])) == 1


def test_nuts_max_tree_depth_caps_work():
    t = orc.TargetSpec(orc.TARGET_DENSE, 16, prec=synth.dense_gaussian_precision(16))
    s = orc.make_settings(seed=3, n_burnin=30, n_keep=30, step=0.01, n_adapt=0, max_depth=3)
    _, info = orc.run_chain(orc.ALGO_NUTS, t, np.zeros(16) + 0.3, s, traces=True)
    assert info["depth"].max() <= 3 and info["leaps"].max() <= 7


def test_nuts_initial_step_size_only_doubles():
    """nuts.ipp:70-89: the loop test is un-signed, so epsilon is 1 or a power of two >= 1."""
    for scale in (1.0, 1e-4, 1e4):
        t = orc.TargetSpec(orc.TARGET_DIAG, 2, prec=np.array([scale, scale]))
        s = orc.make_settings(seed=3, n_burnin=0, n_keep=1, step=1.0, n_adapt=0)
        _, info = orc.run_chain(orc.ALGO_NUTS, t, np.array([0.1, -0.2]), s, traces=True)
        # n_adapt=0 -> step_size = epsilon_bar after the first draw; the first draw used the searched one
        e0 = info["eps_trace"][0]
        assert e0 >= 1.0 and np.log2(e0) == int(np.log2(e0))


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def test_hmc_first_draw_against_a_hand_rolled_trajectory():
    """One HMC draw of a 4-d dense Gaussian recomputed in Python with the stated arithmetic."""
    d, L, eps = 4, 3, 0.25
    P = synth.dense_gaussian_precision(d, seed=1)
    th0 = np.array([0.3, -0.2, 0.5, 0.1])
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P, W=4)
    s = orc.make_settings(seed=77, n_burnin=0, n_keep=1, n_leap=L, step=eps, W=4, chain_id=5)
    dr, info = orc.run_chain(orc.ALGO_HMC, t, th0, s, traces=True)

    def matvec(x):
        out = []
        for i in range(d):
            acc = 0.0
            for k in range(d):
                acc = _fma(P[i, k], x[k], acc)
            out.append(acc)
        return out

    def dot4(x, y):
        q = [0.0] * 4
        for i in range(d):
            q[i % 4] = _fma(x[i], y[i], q[i % 4])
        return (q[0] + q[2]) + (q[1] + q[3])

    p = list(orc.normal_vec(77, 5, 0, 0, d))
    th = list(th0)
    U0 = 0.5 * dot4(th, matvec(th))
    K0 = dot4(p, p) / 2.0
    for _ in range(L):
        w = matvec(th)
        p = [p[i] + (eps * -w[i]) / 2.0 for i in range(d)]
        th = [th[i] + eps * p[i] for i in range(d)]
        w = matvec(th)
        p = [p[i] + (eps * -w[i]) / 2.0 for i in range(d)]
    U1 = 0.5 * dot4(th, matvec(th))
    K1 = dot4(p, p) / 2.0
    comp = min(0.01, -(U1 + K1) + (U0 + K0))
    z = orc.uniform(77, 5, 0, 0)
    accept = z < orc.math_eval(0, np.array([comp]))[0][0]
    assert bool(info["accept"][0]) == accept
    assert np.array_equal(dr[0], np.array(th) if accept else th0)


def test_leapfrog_energy_error_scales_as_eps_squared():
    """Closed-form property of the integrator: |dH| = O(eps^2) on a Gaussian."""
    d = 8
    P = synth.dense_gaussian_precision(d, seed=2)
    errs = []
    for eps in (0.1, 0.05, 0.025):
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P)
        s = orc.make_settings(seed=1, n_burnin=0, n_keep=200, n_leap=int(round(1.0 / eps)), step=eps)
        _, info = orc.run_chain(orc.ALGO_HMC, t, np.zeros(d) + 0.5, s)
        errs.append(1.0 - info["n_accept"] / 200)
    assert errs[0] >= errs[2]            # smaller steps reject less


def test_dmvnorm_against_scipy():
    from scipy.stats import multivariate_normal
    import ctypes as C
    rng = np.random.default_rng(0)
    d = 5
    A = rng.standard_normal((d, d))
    S = A @ A.T + d * np.eye(d)
    x, mu = rng.standard_normal(d), rng.standard_normal(d)
    got = orc.lib().orc_dmvnorm_log(orc._p(x), orc._p(mu), orc._p(np.ascontiguousarray(S)), C.c_size_t(d), 4)
    assert abs(got - multivariate_normal(mu, S).logpdf(x)) < 1e-12


def test_box_transform_round_trip_and_jacobian():
    import ctypes as C
    lb = np.array([0.0, -np.inf, -1.0, -np.inf])
    ub = np.array([np.inf, 2.0, 3.0, np.inf])
    bt = np.zeros(4, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    orc.lib().orc_determine_bounds_type(1, C.c_size_t(4), orc._p(lb), orc._p(ub), bt.ctypes.data_as(ip))
    assert list(bt) == [2, 3, 4, 1]          # determine_bounds_type.hpp:27-57
    v = np.array([1.5, 0.5, 0.25, -7.0])
    tr, back = np.zeros(4), np.zeros(4)
    orc.lib().orc_transform(orc._p(v), bt.ctypes.data_as(ip), orc._p(lb), orc._p(ub), C.c_size_t(4), orc._p(tr))
    orc.lib().orc_inv_transform(orc._p(tr), bt.ctypes.data_as(ip), orc._p(lb), orc._p(ub), C.c_size_t(4), orc._p(back))
    np.testing.assert_allclose(back, v, rtol=0, atol=1e-14)
    np.testing.assert_allclose(tr, [np.log(1.5), -np.log(1.5), np.log(1.25) - np.log(2.75), -7.0], atol=1e-14)
    lj = orc.lib().orc_log_jacobian(orc._p(tr), bt.ctypes.data_as(ip), orc._p(lb), orc._p(ub), C.c_size_t(4))
    e = np.exp(tr[2])
    assert abs(lj - (tr[0] - tr[1] + np.log(4.0) + tr[2] - 2 * np.log(1 + e))) < 1e-13


def test_bounded_hmc_keeps_draws_inside_the_box():
    """vals_bound path (hmc.cpp:88-91,107-122,134-136,211-218): sample sigma>0 of a scaled Gaussian."""
    t = orc.TargetSpec(orc.TARGET_DIAG, 2, prec=np.array([1.0, 4.0]))
    s = orc.make_settings(seed=9, n_burnin=200, n_keep=400, n_leap=5, step=0.2,
                          lower=np.array([0.0, -np.inf]), upper=np.array([np.inf, np.inf]))
    dr, info = orc.run_chain(orc.ALGO_HMC, t, np.array([1.0, 0.0]), s)
    # The reference kicks with inv_jacobian * grad and no Jacobian-gradient term (hmc.cpp:114-122):
    # a valid (reversible, volume-preserving) but poorly aligned proposal, so acceptance is modest.
    assert (dr[:, 0] > 0).all() and 0.2 < info["n_accept"] / 400 <= 1.0
    assert abs(dr[:, 0].mean() - np.sqrt(2 / np.pi)) < 0.3        # half-normal mean sqrt(2/pi)
    assert abs(dr[:, 1].var() - 0.25) < 0.1


def test_example_flow_posterior_of_normal_data():
    """examples/eigen/hmc_normal.cpp: posterior of (mu, sigma) given N(2, 2^2) data is not a built-in
    target; the same plumbing is exercised on its Laplace-like Gaussian: mean recovered, acc in band."""
    P = np.array([[250.0, 0.0], [0.0, 500.0]])       # n/sigma^2, 2n/sigma^2 at n=1000, sigma=2
    t = orc.TargetSpec(orc.TARGET_DENSE, 2, prec=P)
    s = orc.make_settings(seed=1, n_burnin=2000, n_keep=2000, n_leap=1, step=0.08)
    dr, info = orc.run_chain(orc.ALGO_HMC, t, np.array([1.0, 1.0]), s)
    assert np.abs(dr.mean(0)).max() < 0.15
    assert 0.5 < info["n_accept"] / 2000 <= 1.0


def test_many_chain_harness_layout_and_chain_ids():
    d, C = 5, 7
    P = synth.dense_gaussian_precision(d, seed=4)
    init = synth.initial_states(C, d)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P)
    s = orc.make_settings(seed=3, n_burnin=2, n_keep=6, n_leap=3, step=0.1)
    draws, info = orc.run_many(orc.ALGO_HMC, t, init, s, chain0=100)
    assert draws.shape == (6, d, C)
    for c in (0, 3, 6):
        t1 = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P)
        s1 = orc.make_settings(seed=3, n_burnin=2, n_keep=6, n_leap=3, step=0.1, chain_id=100 + c)
        one, i1 = orc.run_chain(orc.ALGO_HMC, t1, init[c], s1)
        assert np.array_equal(draws[:, :, c], one)
        assert info["n_accept"][c] == i1["n_accept"]


def test_logistic_target_gradient_is_the_derivative_of_the_value():
    X, y = synth.logistic_problem(6, 40)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, 6, X=X, y=y)
    b = np.random.default_rng(0).standard_normal(6) * 0.3
    v, g = t.kernel(b)
    eta = X @ b
    assert abs(v - ((y * eta - np.logaddexp(0, eta)).sum() - 0.5 * b @ b)) < 1e-12
    for i in range(6):
        e = np.zeros(6)
        e[i] = 1e-6
        num = (t.kernel(b + e, False)[0] - t.kernel(b - e, False)[0]) / 2e-6
        assert abs(num - g[i]) < 1e-6


# ---------------------------------------------------------------- RWMH (ref: src/rwmh.cpp:30-175)
def test_rwmh_samples_a_gaussian_and_matches_a_numpy_restatement():
    d = 4
    prec = synth.dense_gaussian_precision(d, seed=3)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=1)
    cov = np.linalg.inv(prec)
    s = orc.make_settings(seed=11, n_burnin=200, n_keep=3000, step=0.9, precond=cov, W=1)
    draws, info = orc.run_chain(orc.ALGO_RWMH, t, np.zeros(d), s, traces=True)
    assert 0.15 < info["n_accept"] / 3000 < 0.85
    emp = np.cov(draws.T)
    assert np.allclose(emp, cov, atol=0.35 * np.abs(cov).max())
    # numpy restatement of the loop for the first draws (same normals / uniforms, same accept decisions)
    L = 0.9 * np.linalg.cholesky(cov)
    x, lp = np.zeros(d), 0.0
    for k in range(30):
        z = orc.normal_vec(11, 0, k, 0, d)
        y = x + L @ z
        lpy = -0.5 * y @ prec @ y
        if orc.uniform(11, 0, k, 0) < np.exp(min(0.0, lpy - lp)):
            x, lp = y, lpy
            assert info["accept"][k] == 1
        else:
            assert info["accept"][k] == 0


def test_rwmh_bounded_draws_stay_inside_the_box():
    d = 5
    lb = np.array([-1.0, -np.inf, 0.0, -np.inf, -2.0]); ub = np.array([1.0, 2.0, np.inf, np.inf, 3.0])
    t = orc.TargetSpec(orc.TARGET_ISO, d, W=4)
    s = orc.make_settings(seed=5, n_burnin=50, n_keep=400, step=0.5, W=4, lower=lb, upper=ub)
    draws, info = orc.run_chain(orc.ALGO_RWMH, t, np.array([0.1, 0.2, 0.5, 0.0, 1.0]), s)
    assert (draws >= lb).all() and (draws <= ub).all() and 0 < info["n_accept"] < 400


# ---------------------------------------------------------------- RM-HMC (src/rmhmc.cpp) and the d = 2 normal model

def _normal_model(n=200, seed=0):
    rng = np.random.default_rng(seed)
    x = 2.0 + 2.0 * rng.standard_normal(n)
    return x, orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)


def test_normal_model_kernel_matches_the_example_formula_and_its_gradient():
    x, t = _normal_model()
    v = np.array([1.7, 2.3])
    val, g = t.kernel(v)
    n = x.size
    want = -n * (0.5 * np.log(2 * np.pi) + np.log(v[1])) - ((x - v[0]) ** 2).sum() / (2 * v[1] ** 2)   # rmhmc_normal.cpp:56
    assert abs(val - want) < 1e-10 * abs(want)
    h = 1e-6
    for i in range(2):
        e = np.zeros(2); e[i] = h
        fd = (t.kernel(v + e, want_grad=False)[0] - t.kernel(v - e, want_grad=False)[0]) / (2 * h)
        assert abs(fd - g[i]) < 1e-5 * max(1.0, abs(g[i]))


def test_normal_model_tensor_is_the_fisher_information_and_its_derivative():
    import ctypes as C
    x, t = _normal_model(n=50)
    lib = orc.lib()

    def tensor(v, want_deriv=True):
        G = np.zeros((2, 2)); dG = np.zeros((2, 2, 2))
        vv = np.ascontiguousarray(v, dtype=np.float64)
        lib.orc_target_tensor(vv.ctypes.data_as(C.POINTER(C.c_double)), G.ctypes.data_as(C.POINTER(C.c_double)),
                              dG.ctypes.data_as(C.POINTER(C.c_double)) if want_deriv else None, C.byref(t.c))
        return G, dG

    v = np.array([0.3, 1.9])
    G, dG = tensor(v)
    assert np.allclose(G, np.diag([50 / v[1] ** 2, 100 / v[1] ** 2]), rtol=1e-15)     # rmhmc_normal.cpp:89-90
    h = 1e-6
    for i in range(2):
        e = np.zeros(2); e[i] = h
        fd = (tensor(v + e, False)[0] - tensor(v - e, False)[0]) / (2 * h)
        assert np.allclose(fd, dG[i], atol=1e-6 * np.abs(G).max())


def test_rmhmc_work_accounting_determinism_and_chain_ids():
    x, t = _normal_model()
    init = np.array([[2.5, 2.4], [1.8, 2.9], [2.2, 2.0]])
    s = orc.make_settings(seed=3, n_burnin=4, n_keep=30, n_leap=3, step=0.03, n_fp=4, W=1)
    d1, i1 = orc.run_many(orc.ALGO_RMHMC, t, init, s, chain0=10)
    d2, i2 = orc.run_many(orc.ALGO_RMHMC, t, init, s, chain0=10)
    assert np.array_equal(d1, d2) and np.isfinite(d1).all()
    assert (i1["n_leap"] == 34 * 3).all()                                  # one count per leapfrog step (rmhmc.cpp:215)
    assert 0 < i1["n_accept"].sum() <= 3 * 30
    # chain c of a many-chain run is the single-chain run with chain_id = chain0 + c
    s1 = orc.make_settings(seed=3, n_burnin=4, n_keep=30, n_leap=3, step=0.03, n_fp=4, W=1, chain_id=11)
    single, _ = orc.run_chain(orc.ALGO_RMHMC, t, init[1], s1)
    assert np.array_equal(single, d1[:, :, 1])
    # callbacks per leapfrog: n_fp + 1 gradient calls; per draw one value call (+ one at setup)
    t.c.n_grad_calls = 0; t.c.n_value_calls = 0
    orc.run_chain(orc.ALGO_RMHMC, t, init[0], s1)
    assert t.c.n_grad_calls == 34 * 3 * 5 and t.c.n_value_calls == 34 + 1


def test_rmhmc_with_a_constant_metric_and_no_fixed_point_drift_is_a_valid_sampler():
    # GAUSS target, tensor = precision (constant, zero derivative): the fixed-point loops converge at once.  The reference adds
    # the momentum increment with the opposite sign of Hamilton's equations (rmhmc.cpp:116,139,145), which keeps the map
    # reversible and volume preserving: the chain still targets the right distribution, with a lower acceptance rate.
    d = 3
    P = synth.dense_gaussian_precision(d, seed=2)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P, W=1)
    s = orc.make_settings(seed=8, n_burnin=500, n_keep=20000, n_leap=1, step=0.6, n_fp=3, W=1)
    draws, info = orc.run_chain(orc.ALGO_RMHMC, t, np.zeros(d), s)
    assert 0.2 < info["n_accept"] / 20000 < 0.6
    cov = np.cov(draws.T)
    assert np.allclose(cov, np.linalg.inv(P), atol=0.15)


def test_rmhmc_bounded_draws_stay_inside_the_box():
    x, t = _normal_model(n=80, seed=4)
    lb, ub = np.array([-1.0, 0.5]), np.array([5.0, 6.0])
    s = orc.make_settings(seed=2, n_burnin=5, n_keep=60, n_leap=2, step=0.03, n_fp=3, W=1, lower=lb, upper=ub)
    draws, info = orc.run_chain(orc.ALGO_RMHMC, t, np.array([2.0, 2.0]), s)
    assert (draws > lb).all() and (draws < ub).all() and info["n_accept"] > 0


# ---------------------------------------------------------------- CPU baseline Mode B
@pytest.mark.parametrize("kind,d,precond", [("dense", 24, None), ("dense", 24, "diag"), ("dense", 9, "dense"), ("diag", 40, None),
                                            ("iso", 5, None)])
def test_hmc_mode_b_reproduces_mode_a_bit_for_bit(kind, d, precond):
    """Mode B (gradient reuse, value of the last gradient call, no identity mat-vecs, axpy-form mat-vec from the transposed
    precision, no allocation in the loop) is the same arithmetic as the reference-faithful Mode A: identical draws."""
    rng = np.random.default_rng(5)
    prec = {"dense": synth.dense_gaussian_precision(d, seed=7), "diag": synth.ill_conditioned_diag(d, 30.0), "iso": None}[kind]
    k = {"dense": orc.TARGET_DENSE, "diag": orc.TARGET_DIAG, "iso": orc.TARGET_ISO}[kind]
    M = None
    if precond == "diag":
        M = np.diag(np.linspace(0.5, 2.0, d))
    elif precond == "dense":
        A = rng.standard_normal((d, d)); M = A @ A.T / d + np.eye(d)
    init = synth.initial_states(6, d, seed=2)
    for W in (1, 4):
        t = orc.TargetSpec(k, d, prec=prec, W=W)
        out = []
        for mode in (0, 1):
            s = orc.make_settings(seed=31, n_burnin=4, n_keep=9, n_leap=5, step=0.11, W=W, precond=M, work_mode=mode)
            out.append(orc.run_many(orc.ALGO_HMC, t, init, s, chain0=3, n_threads=2))
        assert np.array_equal(out[0][0], out[1][0])
        assert np.array_equal(out[0][1]["n_accept"], out[1][1]["n_accept"]) and np.array_equal(out[0][1]["n_leap"], out[1][1]["n_leap"])
        assert 0 < out[0][1]["n_accept"].sum()


def test_mode_b_makes_one_gradient_call_per_leapfrog_step():
    d, L, n = 6, 4, 7
    t = orc.TargetSpec(orc.TARGET_ISO, d, W=1)
    s = orc.make_settings(seed=2, n_burnin=0, n_keep=n, n_leap=L, step=0.2, W=1, work_mode=1)
    orc.run_chain(orc.ALGO_HMC, t, np.ones(d), s)
    assert (t.c.n_grad_calls, t.c.n_value_calls) == (n * L + 1, 0)          # Mode A: 2 n L gradient + n + 1 value calls


def test_nuts_adaptation_amplifies_a_change_of_summation_order_but_a_fixed_step_size_does_not():
    """CPU only, oracle against itself: with a fixed step size the dot-product order (W = 1 vs W = 4) reaches a NUTS draw only
    through decisions -- none flips here, the draws are identical -- while inside the dual-averaging window the same change
    re-enters through epsilon and grows (why tests/test_gpu_reference_order.py bounds, not pins, the adapting run)."""
    d = 32
    prec = synth.dense_gaussian_precision(d, seed=7)
    init = synth.initial_states(4, d, seed=3)
    out = {}
    for adapt in (0, 60):
        for W in (1, 4):
            t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=W)
            s = orc.make_settings(seed=4, n_burnin=0, n_keep=120, n_adapt=adapt, step=0.2 if adapt == 0 else 1.0, W=W)
            out[adapt, W] = orc.run_many(orc.ALGO_NUTS, t, init, s, n_threads=2)
    assert np.array_equal(out[0, 1][0], out[0, 4][0]) and np.array_equal(out[0, 1][1]["n_leap"], out[0, 4][1]["n_leap"])
    a, b = out[60, 1][0], out[60, 4][0]
    rel = np.sqrt(((a - b) ** 2).sum(axis=1)) / np.sqrt((a ** 2).sum(axis=1))
    assert rel[:5].max() < 1e-11 and 1e-12 < rel.max() < 1e-1
