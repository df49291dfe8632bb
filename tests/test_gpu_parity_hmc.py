"""GPU: HIP path vs the CPU oracle through the C ABI -- bit-exact (fp64, identical Philox streams)."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _fma(a, b, c):
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def test_native_library_is_loaded_and_sees_a_gpu():
    assert mcmc_amd.lib().mi_mcmc_device_count() >= 1


def test_mfma_f64_accumulates_k_in_order_as_an_fma_chain():
    """The oracle's mat-vec order (sequential fma chain, k ascending) is what the matrix core does."""
    rng = np.random.default_rng(0)
    for _ in range(4):
        A = rng.standard_normal((16, 4)) * 10.0 ** rng.integers(-3, 4, (16, 4))
        B = rng.standard_normal((4, 16)) * 10.0 ** rng.integers(-3, 4, (4, 16))
        C0 = rng.standard_normal((16, 16))
        D = mcmc_amd.probe_mfma(A, B, C0)
        ref = np.empty((16, 16))
        for i in range(16):
            for j in range(16):
                acc = C0[i, j]
                for k in range(4):
                    acc = _fma(A[i, k], B[k, j], acc)
                ref[i, j] = acc
        assert np.array_equal(D, ref)


def test_device_math_matches_oracle_bitwise():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-1, 1, 20000), [0.0, 0.01, -0.0, 710, -800]])
    assert np.array_equal(mcmc_amd.probe_math(0, x)[0], orc.math_eval(0, x)[0])
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0, 2, 20000), [1.0, 5e-324, 2.0 ** -53]])
    assert np.array_equal(mcmc_amd.probe_math(1, x)[0], orc.math_eval(1, x)[0])
    u = np.concatenate([rng.random(40000), [0.0, 0.125, 0.25, 0.5, 0.999999999]])
    gs, gc = mcmc_amd.probe_math(2, u)
    os_, oc = orc.math_eval(2, u)
    assert np.array_equal(gs, os_) and np.array_equal(gc, oc)
    x = rng.uniform(-40, 40, 20000)
    assert np.array_equal(mcmc_amd.probe_math(3, x)[0], orc.math_eval(3, x)[0])
    assert np.array_equal(mcmc_amd.probe_math(4, x)[0], orc.math_eval(4, x)[0])


def test_device_rng_matches_oracle_bitwise():
    for seed, chain, draw, stream, d in [(1, 0, 0, 0, 128), (2 ** 40 + 7, 2 ** 33 + 5, 17, 0, 37),
                                         (99, 65535, 199, 2, 1024), (5, 3, 2, 0, 3)]:
        assert np.array_equal(mcmc_amd.probe_normals(seed, chain, draw, stream, d),
                              orc.normal_vec(seed, chain, draw, stream, d))
    for args in [(1, 0, 0, 0), (7, 123456, 42, 3), (2 ** 63 + 1, 2 ** 35, 2 ** 31, 9)]:
        assert mcmc_amd.probe_uniform(*args) == orc.uniform(*args)


def _oracle_many(kind_orc, d, init, st, prec=None, chain0=0):
    t = orc.TargetSpec(kind_orc, d, prec=prec, W=4)
    s = orc.make_settings(seed=int(st.rng_seed_value), n_burnin=int(st.n_burnin_draws),
                          n_keep=int(st.n_keep_draws), n_leap=int(st.n_leap_steps),
                          step=float(st.step_size), W=4)
    return orc.run_many(orc.ALGO_HMC, t, init, s, chain0=chain0)


CASES = [
    # kind,            d,   C,   L,  eps,  burn, keep
    ("dense", 128, 64, 16, 0.05, 5, 12),     # BASELINE config 2 shape, one workgroup
    ("dense", 128, 37, 4, 0.10, 3, 6),       # ragged chain count (partial wave, dead lanes)
    ("dense", 8, 16, 5, 0.20, 5, 20),        # SURVEY 8(c) golden shape "dense Gaussian d=8"
    ("dense", 100, 130, 3, 0.05, 2, 5),      # d not a multiple of 16, C not a multiple of 64
    ("dense", 33, 17, 2, 0.10, 0, 4),        # odd d, no burn-in
    ("iso", 3, 5, 10, 0.20, 10, 30),         # BASELINE config 1 shape (3-D isotropic Gaussian)
    ("diag", 50, 48, 8, 0.02, 4, 8),
    ("dense", 64, 256, 1, 0.9, 2, 10),       # large step: many rejections exercise the reload path
]


@pytest.mark.parametrize("kind,d,C,L,eps,burn,keep", CASES)
def test_hmc_draws_bit_exact_vs_oracle(kind, d, C, L, eps, burn, keep):
    init = synth.initial_states(C, d, seed=11)
    prec, k_gpu, k_orc = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    if kind == "dense":
        prec, k_gpu, k_orc = synth.dense_gaussian_precision(d, seed=5), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
    elif kind == "diag":
        prec, k_gpu, k_orc = synth.ill_conditioned_diag(d, 100.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    st = mcmc_amd.default_settings(rng_seed_value=1234, n_burnin_draws=burn, n_keep_draws=keep,
                                   n_leap_steps=L, step_size=eps)
    g_draws, g = mcmc_amd.hmc(k_gpu, init, st, prec=prec, chain0=1000)
    o_draws, o = _oracle_many(k_orc, d, init, st, prec=prec, chain0=1000)
    assert np.array_equal(g["n_accept"], o["n_accept"])          # bit-exact accept decisions
    assert np.array_equal(g_draws, o_draws)                      # bit-exact draws
    rel = np.linalg.norm(g_draws - o_draws) / np.linalg.norm(o_draws)
    assert rel <= 1e-9                                           # BASELINE.json tolerance (trivially)
    assert np.array_equal(g["n_leap"], np.full(C, (burn + keep) * L, dtype=np.uint64))
    assert np.array_equal(g["theta"].T, o_draws[-1].T)           # state out = last kept draw


def test_rejections_happen_and_match():
    d, C = 64, 256
    init = synth.initial_states(C, d, seed=11)
    prec = synth.dense_gaussian_precision(d, seed=5)
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=0, n_keep_draws=20, n_leap_steps=1, step_size=0.9)
    _, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert 0 < g["n_accept"].sum() < 20 * C


def test_results_do_not_depend_on_sharding():
    """Chains are keyed by global id: [0,C) in one call == two half shards with chain0 offsets."""
    d, C = 128, 192
    init = synth.initial_states(C, d, seed=3)
    prec = synth.dense_gaussian_precision(d)
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=8, step_size=0.05)
    full, gf = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=0)
    a, ga = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init[:80], st, prec=prec, chain0=0)
    b, gb = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init[80:], st, prec=prec, chain0=80)
    assert np.array_equal(full, np.concatenate([a, b], axis=2))
    assert np.array_equal(gf["n_accept"], np.concatenate([ga["n_accept"], gb["n_accept"]]))


def test_full_size_config2_subset_parity_and_statistics():
    """BASELINE config 2: d=128 dense Gaussian, 65536 chains, L=16, eps=0.05.
    The oracle re-runs a sample of the chains (global ids) bit-exactly; the whole population is
    checked through statistics the domain offers."""
    d, C = 128, 65536
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=3)
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=20, n_keep_draws=10,
                                   n_leap_steps=16, step_size=0.05)
    draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert draws.shape == (10, d, C) and np.isfinite(draws).all()
    pick = np.array([0, 1, 15, 16, 63, 64, 4095, 4096, 32767, 40000, 65534, 65535])
    for c in pick:
        o_draws, o = _oracle_many(orc.TARGET_DENSE, d, init[c:c + 1], st, prec=prec, chain0=int(c))
        assert np.array_equal(draws[:, :, c], o_draws[:, :, 0])
        assert g["n_accept"][c] == o["n_accept"][0]
    acc = g["n_accept"].mean() / 10
    assert 0.9 < acc <= 1.0
    # population covariance of the last draw approaches P^-1 (30 draws x 16 steps from N(0,I) starts):
    last = draws[-1]                                   # [d, C]
    cov = last @ last.T / C
    want = np.linalg.inv(prec)
    assert np.abs(cov - want).max() < 0.1
    assert abs(np.trace(cov) / np.trace(want) - 1) < 0.05


def test_unsupported_requests_fail_loudly():
    """what no device path implements is refused with a status and a reason, never approximated (d = 300 itself now runs, on the
    literal kernel: tests/test_gpu_literal_paths.py)"""
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1, max_tree_depth=40)
    with pytest.raises(mcmc_amd.MiMcmcError) as e:
        mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, np.zeros((4, 300)), st, prec=np.eye(300))
    assert e.value.code == mcmc_amd.MI_ERR_UNSUPPORTED and "max_tree_depth" in str(e.value)


# ---------------------------------------------------------------- elementwise kernel (separable targets, any d)
DIAG_CASES = [
    # kind, d,    C,   L,  eps,    burn, keep
    ("diag", 300, 100, 6, 0.01, 3, 5),        # d > 128 -> lane-per-chain kernel
    ("iso", 1024, 33, 4, 0.05, 2, 3),         # BASELINE config 5 dimension
    ("diag", 1024, 300, 32, 0.005, 1, 2),     # config 5 settings (eps 0.005, L 32, cond 1e4)
    ("diag", 7, 10, 5, 0.10, 4, 9),           # ragged: d not a multiple of 8 (forced onto the diag kernel)
]


@pytest.mark.parametrize("lanes", [1, 4])           # one lane per chain (many chains) / four lanes per chain (fewer chains)
@pytest.mark.parametrize("kind,d,C,L,eps,burn,keep", DIAG_CASES)
def test_hmc_elementwise_kernel_bit_exact_vs_oracle(kind, d, C, L, eps, burn, keep, lanes):
    hint = mcmc_amd.KERNEL_ELEMENTWISE_4LANE if lanes == 4 else mcmc_amd.KERNEL_ELEMENTWISE_1LANE
    init = synth.initial_states(C, d, seed=12)
    prec, k_gpu, k_orc = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
    if kind == "diag":
        prec, k_gpu, k_orc = synth.ill_conditioned_diag(d, 1.0e4), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
    st = mcmc_amd.default_settings(rng_seed_value=55, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
    g_draws, g = mcmc_amd.hmc(k_gpu, init, st, prec=prec, chain0=9, kernel_hint=hint)
    o_draws, o = _oracle_many(k_orc, d, init, st, prec=prec, chain0=9)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    assert np.array_equal(g["theta"], o_draws[-1])
    # draws discarded: same final state
    t = mcmc_amd.make_target(k_gpu, d, prec=prec, kernel_hint=hint)
    theta = np.ascontiguousarray(init.T)
    c = mcmc_amd.make_chains(theta, C, chain0=9)
    mcmc_amd.run("hmc", t, st, c)
    assert np.array_equal(theta, o_draws[-1])


def test_both_hmc_kernels_agree_bitwise_on_a_diagonal_target():
    d, C = 96, 80
    init = synth.initial_states(C, d, seed=12)
    prec = synth.ill_conditioned_diag(d, 100.0)
    st = mcmc_amd.default_settings(rng_seed_value=8, n_burnin_draws=3, n_keep_draws=6, n_leap_steps=7, step_size=0.05)
    a, ga = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec)          # MFMA kernel
    for hint in (mcmc_amd.KERNEL_ELEMENTWISE_1LANE, mcmc_amd.KERNEL_ELEMENTWISE_4LANE):
        b, gb = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec, kernel_hint=hint)          # elementwise kernels
        assert np.array_equal(a, b) and np.array_equal(ga["n_accept"], gb["n_accept"])


# ---------------------------------------------------------------- edge cases
@pytest.mark.parametrize("d,C,L,burn,keep", [
    (1, 1, 3, 2, 4),        # smallest problem: one chain, one dimension
    (128, 1, 2, 0, 1),      # a single chain in a 16-chain wave
    (16, 16, 0, 1, 3),      # n_leap_steps = 0: proposal = current state, still draws momentum and a uniform
    (8, 5, 4, 3, 0),        # n_keep_draws = 0: burn-in only, draws_out empty, state still advances
])
def test_hmc_edge_cases_match_oracle(d, C, L, burn, keep):
    init = synth.initial_states(C, d, seed=13)
    prec = synth.dense_gaussian_precision(d, seed=5)
    st = mcmc_amd.default_settings(rng_seed_value=31, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.1)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    o_draws, o = _oracle_many(orc.TARGET_DENSE, d, init, st, prec=prec)
    assert g_draws.shape == (keep, d, C)
    assert np.array_equal(g_draws, o_draws) and np.array_equal(g["n_accept"], o["n_accept"])
    if keep == 0:       # final state = oracle state after the burn-in (re-run the oracle keeping the last burn-in draw)
        st2 = mcmc_amd.default_settings(rng_seed_value=31, n_burnin_draws=burn - 1, n_keep_draws=1, n_leap_steps=L, step_size=0.1)
        o2, _ = _oracle_many(orc.TARGET_DENSE, d, init, st2, prec=prec)
        assert np.array_equal(g["theta"], o2[-1])


def test_non_finite_energy_is_rejected_like_the_reference():
    """hmc.cpp:180-182: a non-finite proposal energy becomes +inf and the move is rejected."""
    d, C = 4, 16
    init = np.full((C, d), 1e154)                       # theta^T P theta overflows -> U = +inf from the start
    prec = np.eye(d) * 4.0
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=0, n_keep_draws=3, n_leap_steps=2, step_size=0.5)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    o_draws, o = _oracle_many(orc.TARGET_DENSE, d, init, st, prec=prec)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True)


# ---------------------------------------------------------------- box constraints (settings.vals_bound), SURVEY 8(f-1)
def _bounds(d, seed=0):
    """A mix of the four bounds types of determine_bounds_type.hpp:27-57."""
    rng = np.random.default_rng(seed)
    kind = rng.integers(1, 5, d)
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf)
    ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    return lb, ub


@pytest.mark.parametrize("d,C,L,eps,burn,keep", [(6, 16, 4, 0.05, 4, 12), (128, 48, 3, 0.02, 2, 5), (37, 20, 2, 0.05, 0, 6)])
def test_bounded_hmc_bit_exact_vs_oracle(d, C, L, eps, burn, keep):
    lb, ub = _bounds(d, seed=d)
    prec = synth.dense_gaussian_precision(d, seed=5)
    init = np.clip(synth.initial_states(C, d, seed=14) * 0.3, -1.0, 1.5)     # inside every box
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L,
                                   step_size=eps, vals_bound=1, lower_bounds=lb, upper_bounds=ub)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=77, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, lower=lb, upper=ub)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s, chain0=4)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)
    inside = (g_draws >= lb[None, :, None]) & (g_draws <= ub[None, :, None])
    assert inside.all()                                  # draws are reported in the constrained space


# ---------------------------------------------------------------- precond_mat (diagonal), SURVEY 8(f-2)
@pytest.mark.parametrize("d,C,L,eps,bounded", [(8, 16, 5, 0.1, False), (128, 40, 3, 0.03, False), (20, 24, 4, 0.05, True)])
def test_diagonal_precond_hmc_bit_exact_vs_oracle(d, C, L, eps, bounded):
    prec = synth.dense_gaussian_precision(d, seed=5)
    M = np.diag(1.0 / np.diag(prec) * np.linspace(0.5, 2.0, d))        # a diagonal mass matrix
    init = np.clip(synth.initial_states(C, d, seed=15) * 0.3, -1.0, 1.5)
    kw, okw = {}, {}
    if bounded:
        lb, ub = _bounds(d, seed=3)
        kw = dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub)
        okw = dict(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=3, n_keep_draws=7, n_leap_steps=L, step_size=eps,
                                   precond_mat=M, **kw)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=5, n_burnin=3, n_keep=7, n_leap=L, step=eps, W=4, precond=M, **okw)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    return A @ A.T + np.diag(rng.uniform(0.5, 1.5, d))


@pytest.mark.parametrize("d,C,L,eps,bounded", [(8, 16, 5, 0.1, False), (37, 40, 3, 0.05, False), (64, 33, 4, 0.04, False), (20, 24, 4, 0.05, True),
                                               (128, 70, 16, 0.03, False),      # the BASELINE dimension: INV(M), CHOL(M) read from L2 in fragment order
                                               (100, 33, 4, 0.03, True), (65, 16, 3, 0.04, False)])
def test_dense_precond_hmc_bit_exact_vs_oracle(d, C, L, eps, bounded):
    """precond_mat dense: p = L z, theta += eps Minv p, K = p.Minv p / 2 as MFMA mat-vecs; INV / CHOL on the host."""
    prec = synth.dense_gaussian_precision(d, seed=5)
    M = _spd(d, seed=d)
    init = np.clip(synth.initial_states(C, d, seed=16) * 0.3, -1.0, 1.5)
    kw, okw = {}, {}
    if bounded:
        lb, ub = _bounds(d, seed=4)
        kw = dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub)
        okw = dict(lower=lb, upper=ub)
    st = mcmc_amd.default_settings(rng_seed_value=8, n_burnin_draws=3, n_keep_draws=7, n_leap_steps=L, step_size=eps,
                                   precond_mat=M, **kw)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=2)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=8, n_burnin=3, n_keep=7, n_leap=L, step=eps, W=4, precond=M, **okw)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s, chain0=2)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert 0 < g["n_accept"].sum()


def test_dense_precond_beyond_128_dims_runs_literally():
    """three d x d fragment sets beyond d = 128 fit neither LDS nor the tiled kernels' registers: the literal kernel runs it"""
    d, C = 130, 6
    M = _spd(d, seed=1)
    init = synth.initial_states(C, d, seed=2)
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=1, n_keep_draws=3, n_leap_steps=2, step_size=0.1, precond_mat=M)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_ISO, init, st)
    assert mcmc_amd.last_kernel() == "literal_kernel<0>"
    s = orc.make_settings(seed=3, n_burnin=1, n_keep=3, n_leap=2, step=0.1, W=4, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_ISO, d, W=4), init, s)
    assert np.array_equal(g_draws, o_draws) and np.array_equal(g["n_accept"], o["n_accept"])


# ---------------------------------------------------------------- HMC on the logistic-regression target
HMC_LOGIT_CASES = [
    # d, N, C, eps, L, burn, keep
    (5, 40, 16, 0.05, 8, 5, 20),        # SURVEY 8(c) golden shape "logistic d=5"
    (64, 100, 33, 0.02, 4, 2, 6),
    (100, 37, 20, 0.03, 5, 0, 8),       # ragged rows and dims
    (512, 1024, 32, 0.01, 3, 1, 3),     # config 3 dimensions
    (300, 64, 17, 0.30, 6, 0, 10),      # big step: rejections and non-finite energies
    (16, 50, 16, 0.05, 0, 0, 4),        # n_leap_steps = 0
]


@pytest.mark.parametrize("d,N,C,eps,L,burn,keep", HMC_LOGIT_CASES)
def test_hmc_logistic_bit_exact_vs_oracle(d, N, C, eps, L, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps, n_leap_steps=L)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=5)
    dq = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=dq, eta_chains=2)
    s = orc.make_settings(seed=77, n_burnin=burn, n_keep=keep, step=eps, n_leap=L, W=4, hoist=1, blocks=4, block_size=dq)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s, chain0=5)
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g_draws, o_draws)


# ---------------------------------------------------------------- diagonal precond_mat on the elementwise kernel (any d)
@pytest.mark.parametrize("lanes", [1, 4])
@pytest.mark.parametrize("kind,d,C,L,eps", [("diag", 200, 70, 6, 0.2), ("iso", 1024, 33, 4, 0.08), ("diag", 129, 300, 3, 0.1)])
def test_diagonal_precond_beyond_128_dims_bit_exact_vs_oracle(kind, d, C, L, eps, lanes):
    hint = mcmc_amd.KERNEL_ELEMENTWISE_4LANE if lanes == 4 else mcmc_amd.KERNEL_ELEMENTWISE_1LANE
    prec = synth.ill_conditioned_diag(d, 50.0) if kind == "diag" else None
    M = np.diag((prec if prec is not None else np.ones(d)) * np.linspace(0.7, 1.4, d))   # mass matrix ~ the precision: every dimension at unit frequency
    init = synth.initial_states(C, d, seed=18) * 0.5
    st = mcmc_amd.default_settings(rng_seed_value=12, n_burnin_draws=2, n_keep_draws=6, n_leap_steps=L, step_size=eps, precond_mat=M)
    k_gpu, k_orc = (mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG) if kind == "diag" else (mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)
    g_draws, g = mcmc_amd.hmc(k_gpu, init, st, prec=prec, chain0=6, kernel_hint=hint)
    t = orc.TargetSpec(k_orc, d, prec=prec, W=4)
    s = orc.make_settings(seed=12, n_burnin=2, n_keep=6, n_leap=L, step=eps, W=4, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s, chain0=6)
    assert np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g_draws, o_draws)
    assert g["n_accept"].sum() > 0


# ---------------------------------------------------------------- launch shapes of the plain kernel for few chains (strong scaling)
@pytest.mark.parametrize("C", [1, 16, 33, 200])
@pytest.mark.parametrize("d", [128, 100, 65])
def test_few_chain_launch_shapes_give_the_bits_of_the_default_kernel(d, C):
    """One wave per SIMD, two and four waves per 16-chain tile (hmc_split.hpp: row blocks of the mat-vec per wave, theta exchanged
    through LDS, dot-product chains relayed in slice order) against the default shape and the oracle."""
    prec = synth.dense_gaussian_precision(d, seed=9)
    init = synth.initial_states(C, d, seed=14)
    st = mcmc_amd.default_settings(rng_seed_value=41, n_burnin_draws=3, n_keep_draws=7, n_leap_steps=5, step_size=0.07)
    ref, gr = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=77, kernel_hint=mcmc_amd.KERNEL_HMC_TWO_WAVES_PER_SIMD)
    o_draws, o = _oracle_many(orc.TARGET_DENSE, d, init, st, prec=prec, chain0=77)
    assert np.array_equal(ref, o_draws) and np.array_equal(gr["n_accept"], o["n_accept"])
    for hint in (mcmc_amd.KERNEL_HMC_ONE_WAVE_PER_SIMD, mcmc_amd.KERNEL_HMC_SPLIT2, mcmc_amd.KERNEL_HMC_SPLIT4, mcmc_amd.KERNEL_HMC_SPLIT4_TWO_WAVES,
                 mcmc_amd.KERNEL_AUTO):
        got, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=77, kernel_hint=hint)
        assert np.array_equal(got, ref), hint
        assert np.array_equal(g["n_accept"], gr["n_accept"]) and np.array_equal(g["theta"], gr["theta"]) and np.array_equal(g["n_leap"], gr["n_leap"])


def test_split_kernel_with_zero_leapfrog_steps_and_discarded_draws():
    d, C = 128, 40
    prec = synth.dense_gaussian_precision(d, seed=9)
    init = synth.initial_states(C, d, seed=14)
    for L in (0, 1):
        st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=2, n_keep_draws=3, n_leap_steps=L, step_size=0.07)
        ref, gr = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=mcmc_amd.KERNEL_HMC_TWO_WAVES_PER_SIMD)
        for hint in (mcmc_amd.KERNEL_HMC_SPLIT2, mcmc_amd.KERNEL_HMC_SPLIT4, mcmc_amd.KERNEL_HMC_SPLIT4_TWO_WAVES):
            got, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=hint)
            assert np.array_equal(got, ref) and np.array_equal(g["n_accept"], gr["n_accept"])
            _, g2 = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=hint, want_draws=False)
            assert np.array_equal(g2["theta"], gr["theta"])
