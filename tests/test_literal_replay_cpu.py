"""The product's literal replay (mcmc_amd/csrc/literal.hpp), compiled for the HOST, against the oracle -- bit for bit, with the
emphasis on the non-finite regime: step sizes that blow a chain up, initial values that are already
+-inf / NaN, bounds whose Jacobian overflows, every preconditioner form.  The throughput kernels hand exactly such chains to
this code on the GPU (tests/test_gpu_nonfinite.py checks that hand-over); here the replay itself is pinned without a GPU."""
import numpy as np
import pytest

import lit_host
import orc
from mcmc_amd import synth


def _case(rng, algo, tgt, force_nonfinite):
    d = int(rng.choice([1, 2, 3, 5, 8, 12, 17])) if tgt != "logit" else int(rng.choice([3, 9, 17, 40, 70]))
    if algo == "rmhmc": d = min(d, 9)                        # O(d^4) per fixed-point step
    C = int(rng.choice([1, 2, 5]))
    seed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
    burn, keep = int(rng.integers(0, 3)), int(rng.integers(1, 6))
    L = int(rng.integers(0, 5))
    depth, adapt = int(rng.integers(0, 6)), int(rng.integers(0, 4))
    n_fp = int(rng.integers(0, 4))
    if algo == "rmhmc": L = min(L, 2)
    eps = float(rng.choice([0.05, 0.3, 1.5, 40.0, 1e6, 1e160] if force_nonfinite else [0.01, 0.1, 0.5]))
    prec = X = y = None
    if tgt == "dense": prec, ko = synth.dense_gaussian_precision(d, seed=seed % 97), orc.TARGET_DENSE
    elif tgt == "diag": prec, ko = synth.ill_conditioned_diag(d, 50.0), orc.TARGET_DIAG
    elif tgt == "iso": ko = orc.TARGET_ISO
    else:
        N = int(rng.choice([1, 7, 16, 33])); X, y = synth.logistic_problem(d, N, seed=seed % 89); ko = orc.TARGET_LOGISTIC
    init = synth.initial_states(C, d, seed=seed % 1013) * float(rng.choice([0.1, 1.0, 1e3, 1e200] if force_nonfinite else [0.1, 1.0]))
    if force_nonfinite and rng.random() < 0.4:
        bad = rng.choice([np.inf, -np.inf, np.nan])
        init[rng.integers(0, C), rng.integers(0, d)] = bad
    kw, okw = {}, {}
    if tgt != "logit" and rng.random() < 0.6:
        if rng.random() < 0.6:
            kind = rng.integers(1, 5, d)
            lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
            kw.update(lower=lb, upper=ub); okw.update(lower=lb, upper=ub)
            fin = np.isfinite(init)
            init = np.where(fin, np.clip(init, -1.0, 1.5), init)
        if rng.random() < 0.6 and algo != "rmhmc":           # rmhmc has no precond_mat
            M = np.diag(rng.uniform(0.3, 3.0, d))
            if rng.random() < 0.5:
                A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + M
            kw.update(precond=M); okw.update(precond=M)
    tkw = {}
    if tgt == "logit":
        dq = 16 if d <= 64 else 32
        tkw = dict(blocks=4, block_size=dq, eta_chains=2); okw.update(blocks=4, block_size=dq)
    t = orc.TargetSpec(ko, d, prec=prec, X=X, y=y, W=4, **tkw)
    s = orc.make_settings(seed=seed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1, n_adapt=adapt, max_depth=depth, n_fp=n_fp, **okw)
    o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH, "rmhmc": orc.ALGO_RMHMC}[algo], t, init, s, chain0=chain0)
    l_draws, l = lit_host.run(algo, tgt, init, seed, burn, keep, L, eps, prec=prec, X=X, y=y, chain0=chain0, n_adapt=adapt, max_depth=depth, n_fp=n_fp, **kw)
    desc = f"{algo} {tgt} d={d} C={C} eps={eps} L={L} burn={burn} keep={keep} depth={depth} adapt={adapt} opts={sorted(kw)}"
    ok = np.array_equal(l_draws, o_draws, equal_nan=True) and np.array_equal(l["n_accept"], o["n_accept"])
    if algo in ("hmc", "rmhmc"):
        ok = ok and np.array_equal(l["n_leap"], o["n_leap"])
    if algo == "nuts":
        ok = ok and np.array_equal(l["n_leap"], o["n_leap"]) and np.array_equal(l["eps"], o["eps"], equal_nan=True)
    return ok, desc, bool(np.isnan(o_draws).any() or np.isinf(o_draws).any())


@pytest.mark.parametrize("algo", ["hmc", "mala", "nuts", "rwmh", "rmhmc"])
@pytest.mark.parametrize("tgt", ["dense", "iso", "diag", "logit"])
def test_literal_replay_equals_the_oracle_in_the_finite_regime(algo, tgt):
    rng = np.random.default_rng([11, len(algo), len(tgt)])
    for _ in range(12):
        ok, desc, _nf = _case(rng, algo, tgt, force_nonfinite=False)
        assert ok, desc


@pytest.mark.parametrize("algo", ["hmc", "mala", "nuts", "rwmh", "rmhmc"])
@pytest.mark.parametrize("tgt", ["dense", "iso", "diag", "logit"])
def test_literal_replay_equals_the_oracle_in_the_non_finite_regime(algo, tgt):
    rng = np.random.default_rng([12, len(algo), len(tgt)])
    n_nonfinite = 0
    for _ in range(40):
        ok, desc, nf = _case(rng, algo, tgt, force_nonfinite=True)
        assert ok, desc
        n_nonfinite += int(nf)
    assert n_nonfinite >= 8, f"the sweep is meant to reach the non-finite regime ({n_nonfinite} of 40 cases did)"


@pytest.mark.parametrize("algo", ["hmc", "mala", "rwmh", "nuts"])
@pytest.mark.parametrize("d", [129, 200, 256, 300, 512])
def test_literal_dense_gaussian_uses_the_blocked_dot_order_between_128_and_512(algo, d):
    """128 < d <= 512: the dense Gaussian runs on the LDS-streamed kernel (logistic_lds.hpp); its energies are summed over four dimension
    quarters, and the literal replay of its flagged chains has to use that order."""
    rng = np.random.default_rng([13, len(algo), d])
    prec = synth.dense_gaussian_precision(d, seed=d % 97)
    bs = 48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128
    for eps, scale in ((0.05, 1.0), (1e6, 1e3)):
        init = synth.initial_states(2, d, seed=d) * scale
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, blocks=4, block_size=bs)
        s = orc.make_settings(seed=77, n_burnin=1, n_keep=3, n_leap=3, step=eps, W=4, hoist=1, n_adapt=2, max_depth=3, blocks=4, block_size=bs)
        o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH}[algo], t, init, s, chain0=5)
        l_draws, l = lit_host.run(algo, "dense", init, 77, 1, 3, 3, eps, prec=prec, chain0=5, n_adapt=2, max_depth=3)
        assert np.array_equal(l_draws, o_draws, equal_nan=True) and np.array_equal(l["n_accept"], o["n_accept"]), (algo, d, eps)


@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("d", [129, 300])
def test_literal_dense_gaussian_with_a_dense_precond_mat_between_128_and_512(algo, d):
    """The replay behind round 5's DENSEM instantiations of the LDS-streamed kernel (hmc / mala with a dense precond_mat): INV /
    CHOL_LOWER of M (mala: INV and LOG_DET of eps^2 M) from host_linalg.hpp -- the vectorised elimination, the memo (the second step size asks
    for INV(M) again) -- and the blocked dot order, against the oracle."""
    rng = np.random.default_rng([14, len(algo), d])
    prec = synth.dense_gaussian_precision(d, seed=d % 97)
    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.3, 3.0, d))
    D = np.ones(d); D[[0, 7, d // 2]] = 0.02              # D M D: still symmetric positive definite, and columns whose diagonal entry is not the
    M = D[:, None] * M * D[None, :]                        # largest -- the elimination has to pivot
    assert np.abs(M[1:, 0]).max() > M[0, 0]
    bs = 48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128
    for eps, scale in ((0.01, 0.3), (1e6, 1e3), (0.02, 0.3)):
        init = synth.initial_states(2, d, seed=d) * scale
        t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4, blocks=4, block_size=bs)
        s = orc.make_settings(seed=78, n_burnin=1, n_keep=3, n_leap=3, step=eps, W=4, hoist=1, precond=M, blocks=4, block_size=bs)
        o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA}[algo], t, init, s, chain0=5)
        l_draws, l = lit_host.run(algo, "dense", init, 78, 1, 3, 3, eps, prec=prec, chain0=5, precond=M)
        assert np.array_equal(l_draws, o_draws, equal_nan=True) and np.array_equal(l["n_accept"], o["n_accept"]), (algo, d, eps)
        if scale <= 1.0: assert np.isfinite(o_draws).all() and o["n_accept"].sum() > 0, "the matrix is meant to be a usable preconditioner"


# ---- the features round 3 added to the literal kernels, pinned on the CPU as well (host instantiation): per-chain diagonal masses, host
# callbacks as the target (the host branch of the mailbox), the nuts dual-averaging state across a cut

@pytest.mark.parametrize("tgt", ["iso", "dense", "logit"])
def test_literal_hmc_with_per_chain_masses_equals_the_oracle_chain_by_chain(tgt):
    rng = np.random.default_rng([21, len(tgt)])
    d, C = 11, 6
    prec = X = y = None; tkw, okw = {}, {}
    if tgt == "dense": prec, ko = synth.dense_gaussian_precision(d, seed=3), orc.TARGET_DENSE
    elif tgt == "iso": ko = orc.TARGET_ISO
    else:
        X, y = synth.logistic_problem(d, 17, seed=3); ko = orc.TARGET_LOGISTIC
        tkw = dict(blocks=4, block_size=16, eta_chains=2); okw = dict(blocks=4, block_size=16)
    init = synth.initial_states(C, d, seed=5) * 0.5
    init[2, 3] = np.inf                                  # one chain in the non-finite regime
    mass = np.ascontiguousarray(rng.uniform(0.2, 5.0, (d, C)))
    l_draws, l = lit_host.run("hmc", tgt, init, 31, 2, 5, 4, 0.2, prec=prec, X=X, y=y, chain0=7, mass_diag=mass)
    t = orc.TargetSpec(ko, d, prec=prec, X=X, y=y, W=4, **tkw)
    for c in range(C):
        s = orc.make_settings(seed=31, n_burnin=2, n_keep=5, n_leap=4, step=0.2, W=4, hoist=1, precond=np.diag(mass[:, c]), **okw)
        o, info = orc.run_many(orc.ALGO_HMC, t, init[c:c + 1], s, chain0=7 + c)
        assert np.array_equal(l_draws[:, :, c], o[:, :, 0], equal_nan=True) and l["n_accept"][c] == info["n_accept"][0], (tgt, c)


@pytest.mark.parametrize("algo", ["hmc", "mala", "nuts", "rwmh", "rmhmc"])
def test_literal_samplers_with_host_callbacks_as_target_equal_the_oracle(algo):
    """kind = LIT_CALLBACK: every target / tensor evaluation goes through the mailbox -- on the host a direct call of the user's function
    pointers, here the oracle's own target and tensor (so the run must equal the oracle's)."""
    import ctypes as C
    d = 5
    X, y = synth.logistic_problem(d, 20, seed=9)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4)
    kern = C.cast(orc.lib().orc_target_kernel, C.c_void_p).value
    tens = C.cast(orc.lib().orc_target_tensor, C.c_void_p).value
    lb = np.array([-1.5, -np.inf, -np.inf, -2.0, -np.inf]); ub = np.array([2.0, np.inf, 1.8, np.inf, np.inf])
    M = None if algo == "rmhmc" else np.diag([0.5, 1.0, 2.0, 1.5, 0.8])
    init = np.clip(synth.initial_states(2, d, seed=4) * 0.3, -1.0, 1.5)
    l_draws, l = lit_host.run(algo, "callback", init, 17, 2, 6, 3, 0.1, lower=lb, upper=ub, precond=M, n_adapt=2, max_depth=4, n_fp=3,
                              kernel_cb=kern, kernel_data=C.addressof(t.c), tensor_cb=tens, tensor_data=C.addressof(t.c))
    okw = dict(lower=lb, upper=ub) if M is None else dict(lower=lb, upper=ub, precond=M)
    s = orc.make_settings(seed=17, n_burnin=2, n_keep=6, n_leap=3, step=0.1, n_adapt=2, max_depth=4, n_fp=3, W=4, **okw)
    a = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH, "rmhmc": orc.ALGO_RMHMC}[algo]
    o_draws, o = orc.run_many(a, t, init, s)
    assert np.array_equal(l_draws, o_draws, equal_nan=True) and np.array_equal(l["n_accept"], o["n_accept"])


@pytest.mark.parametrize("cut", [2, 7, 8, 11])
def test_literal_nuts_continues_from_a_cut_anywhere_with_its_dual_averaging_state(cut):
    """n_adapt_draws = 8 of 14 draws: cut inside the window, at its end, after it -- bit-identical to the uncut run (host instantiation)"""
    d, C, n_tot, n_adapt = 7, 4, 14, 8
    prec = synth.dense_gaussian_precision(d, seed=2)
    init = synth.initial_states(C, d, seed=6) * 0.5
    kw = dict(prec=prec, chain0=3, n_adapt=n_adapt, max_depth=4)
    w_draws, w = lit_host.run("nuts", "dense", init, 5, 0, n_tot, 0, 0.1, **kw)
    a_draws, a = lit_host.run("nuts", "dense", init, 5, 0, cut, 0, 0.1, **kw)
    b_draws, b = lit_host.run("nuts", "dense", a["theta"], 5, 0, n_tot - cut, 0, 0.1, draw0=cut, step_in=a["eps"], adapt_state_in=a["adapt_state"], **kw)
    assert np.array_equal(np.concatenate([a_draws, b_draws]), w_draws)
    assert np.array_equal(b["eps"], w["eps"]) and np.array_equal(a["n_leap"] + b["n_leap"], w["n_leap"])
