"""GPU: the non-finite regime.  The reference multiplies by its identity / diagonal matrices as DENSE products
(ref: src/hmc.cpp:160,171,184; src/mala.cpp:115-123,155-157; include/mcmc/mala.ipp:52-64), so ONE non-finite entry of a momentum /
gradient / Jacobian turns every other dimension of the chain into NaN, and std::min(0.01, NaN) accepts the result.  The throughput
kernels detect such chains and the literal kernels (mcmc_amd/csrc/literal.hpp) replay them; these tests drive chains non-finite on
every plain path and compare with the oracle bit for bit -- next to healthy chains in the same launch, which must stay untouched."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def _same(g_draws, g, o_draws, o):
    assert np.array_equal(g["n_accept"], o["n_accept"]), "accept counts differ"
    bad = np.argwhere(~((g_draws == o_draws) | (np.isnan(g_draws) & np.isnan(o_draws))))
    assert bad.size == 0, f"draws differ, first at [keep, dim, chain] = {bad[0].tolist()}"


def _poisoned_chains(o_draws):
    """chains whose last kept row holds a non-finite value"""
    return np.flatnonzero((~np.isfinite(o_draws[-1])).any(axis=0))


@pytest.mark.parametrize("hint", [mcmc_amd.KERNEL_ELEMENTWISE_4LANE, mcmc_amd.KERNEL_ELEMENTWISE_1LANE])
@pytest.mark.parametrize("precond", [False, True])
def test_elementwise_hmc_d300_chain_driven_non_finite(hint, precond):
    """hmc_diag{1,4}_kernel (any d): a step size beyond the stability limit of the stiff dimensions lets them overflow inside a
    trajectory; the reference then poisons the STABLE dimensions too.  Chains 0..5 start huge, the rest stay finite."""
    d, C = 300, 70
    prec = synth.ill_conditioned_diag(d, 1.0e4)
    init = synth.initial_states(C, d, seed=5)
    init[:6] *= 1.0e300
    init[7, 11] = np.inf
    kw, okw = {}, {}
    if precond:
        if hint == mcmc_amd.KERNEL_ELEMENTWISE_1LANE: pytest.skip("one case of the preconditioned elementwise kernel is enough")
        M = np.diag(np.random.default_rng(3).uniform(0.5, 2.0, d)); kw["precond_mat"] = M; okw["precond"] = M
    st = mcmc_amd.default_settings(rng_seed_value=11, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=6, step_size=0.5, **kw)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec, kernel_hint=hint if not precond else mcmc_amd.KERNEL_AUTO)
    t = orc.TargetSpec(orc.TARGET_DIAG, d, prec=prec, W=4)
    s = orc.make_settings(seed=11, n_burnin=2, n_keep=4, n_leap=6, step=0.5, W=4, **okw)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s)
    pc = _poisoned_chains(o_draws)
    assert len(pc) >= 6 and len(pc) < C, f"the case is meant to mix poisoned and healthy chains ({len(pc)} poisoned)"
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("hint", [mcmc_amd.KERNEL_AUTO, mcmc_amd.KERNEL_HMC_ONE_WAVE_PER_SIMD, mcmc_amd.KERNEL_HMC_SPLIT2,
                                  mcmc_amd.KERNEL_HMC_SPLIT4_TWO_WAVES, mcmc_amd.KERNEL_HMC_SPLIT4])
def test_plain_mfma_hmc_d128_chain_driven_non_finite(hint):
    """hmc_gauss_mfma_kernel<8, ., false> and the split-tile shapes: chains that start at 1e300 overflow in the first mat-vec."""
    d, C = 128, 70
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=9)
    init[3] *= 1.0e300
    init[20, 5] = -np.inf
    init[40, 127] = np.nan
    init[64] *= 1.0e160       # finite gradient, overflow in the energy only
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=1, n_keep_draws=4, n_leap_steps=3, step_size=0.1)
    g_draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, kernel_hint=hint)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    s = orc.make_settings(seed=4, n_burnin=1, n_keep=4, n_leap=3, step=0.1, W=4)
    o_draws, o = orc.run_many(orc.ALGO_HMC, t, init, s)
    assert len(_poisoned_chains(o_draws)) >= 2
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("d,kind", [(128, "dense"), (40, "dense"), (100, "diag")])
def test_mfma_hmc_with_a_diagonal_precond_mat_alone_non_finite(d, kind):
    """hmc_gauss_mfma_kernel<., 8, false, false, true> (a diagonal precond_mat without bounds runs in the plain kernel's shape):
    the reference's dense `inv_precond_matrix * mntm` poisons every dimension once one is non-finite -- detected through the energies,
    replayed by the literal kernel with the same diagonal matrix."""
    C = 70
    prec = synth.dense_gaussian_precision(d) if kind == "dense" else synth.ill_conditioned_diag(d, 50.0)
    M = np.diag(np.random.default_rng(5).uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=9)
    init[3] *= 1.0e300; init[20, 5] = -np.inf; init[40, d - 1] = np.nan; init[64] *= 1.0e160
    kg, ko = (mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE) if kind == "dense" else (mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG)
    for eps in (0.1, 1.0e6):
        st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=1, n_keep_draws=4, n_leap_steps=3, step_size=eps, precond_mat=M)
        g_draws, g = mcmc_amd.hmc(kg, init, st, prec=prec)
        assert mcmc_amd.last_kernel().endswith("false, false, true>"), mcmc_amd.last_kernel()
        s = orc.make_settings(seed=4, n_burnin=1, n_keep=4, n_leap=3, step=eps, W=4, precond=M)
        o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(ko, d, prec=prec, W=4), init, s)
        assert len(_poisoned_chains(o_draws)) >= 2
        _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("kind", ["iso", "diag"])
def test_plain_mfma_hmc_on_separable_targets_non_finite(kind):
    """ISO / DIAG targets on the MFMA kernel (d <= 128): the oracle's target is element-wise there, the kernel's a mat-vec over a
    diagonal matrix -- they differ only once a value is non-finite, which is when the chain is replayed."""
    d, C = 100, 33
    prec = synth.ill_conditioned_diag(d, 1.0e3) if kind == "diag" else None
    init = synth.initial_states(C, d, seed=2)
    init[1, 0] = np.inf
    init[17] *= 1.0e300
    st = mcmc_amd.default_settings(rng_seed_value=8, n_burnin_draws=0, n_keep_draws=3, n_leap_steps=4, step_size=0.7)
    kg, ko = (mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG) if kind == "diag" else (mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)
    g_draws, g = mcmc_amd.hmc(kg, init, st, prec=prec)
    s = orc.make_settings(seed=8, n_burnin=0, n_keep=3, n_leap=4, step=0.7, W=4)
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(ko, d, prec=prec, W=4), init, s)
    assert len(_poisoned_chains(o_draws)) >= 1
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("variant", ["plain", "diag_precond", "bounded", "bounded_diag_precond"])
def test_mala_gauss_chain_driven_non_finite(variant):
    """mala_gauss_mfma_kernel plain and general: a huge step size throws the proposal far out -- gradients overflow, and in a bounded
    run inv_jacobian_adjust overflows, Gauss-Jordan then pivots on inf / NaN (mala.ipp:52-64)."""
    d, C = 100, 40
    prec = synth.dense_gaussian_precision(d, seed=6)
    init = np.clip(synth.initial_states(C, d, seed=12), -1.0, 1.5)
    kw, okw = {}, {}
    if "bounded" in variant:
        kind = np.random.default_rng(1).integers(1, 5, d)
        lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    else:
        init[2] *= 1.0e300; init[5, 9] = np.inf
    if "precond" in variant:
        M = np.diag(np.random.default_rng(2).uniform(0.3, 3.0, d)); kw["precond_mat"] = M; okw["precond"] = M
    eps = 1.0e3 if "bounded" in variant else 0.3
    st = mcmc_amd.default_settings(rng_seed_value=21, n_burnin_draws=1, n_keep_draws=4, step_size=eps, **kw)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    s = orc.make_settings(seed=21, n_burnin=1, n_keep=4, step=eps, W=4, hoist=1, **okw)
    o_draws, o = orc.run_many(orc.ALGO_MALA, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("d,C", [(20, 33), (100, 8)])
def test_bounded_mala_with_a_dense_preconditioner_runs_on_the_literal_kernel(d, C):
    """ref: src/mala.cpp:152-157, include/mcmc/mala.ipp:45-55 -- INV(eps^2 J(theta') M) per draw: every chain on literal.hpp"""
    prec = synth.dense_gaussian_precision(d, seed=3)
    init = np.clip(synth.initial_states(C, d, seed=4), -1.0, 1.5)
    rng = np.random.default_rng(9)
    kind = rng.integers(1, 5, d)
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.3, 3.0, d))
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=2, n_keep_draws=5, step_size=0.15, vals_bound=1,
                                   lower_bounds=lb, upper_bounds=ub, precond_mat=M)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    s = orc.make_settings(seed=5, n_burnin=2, n_keep=5, step=0.15, W=4, hoist=1, lower=lb, upper=ub, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_MALA, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    assert 0 < o["n_accept"].sum() < 5 * C, "the case is meant to both accept and reject"
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("kind,d", [("diag", 31), ("iso", 47), ("dense", 31), ("dense", 100)])
def test_unbounded_mala_with_a_dense_preconditioner_and_chains_started_non_finite(kind, d):
    """mala_gauss_dense_m_kernel has no replay -- its products ARE the reference's dense products --, so two things have to be literal: an ISO /
    DIAG target's gradient is element-wise (the reference's target function: +-inf stays in its own dimension; the mat-vec over the expanded
    diagonal made every other dimension NaN), and the padding dimensions of a d that does not fill its tiles stay 0 (0 * inf there fed NaN
    back through the next product).  Found by the round-5 random sweep (fuzz_parity seed 4242: diag, d = 31, a chain started at -inf)."""
    C = 20
    prec, kg, ko = {"dense": (synth.dense_gaussian_precision(d, seed=3), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE),
                    "diag": (synth.ill_conditioned_diag(d, 20.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG),
                    "iso": (None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)}[kind]
    rng = np.random.default_rng(d)
    init = synth.initial_states(C, d, seed=4)
    init[3, 6] = -np.inf; init[8, d - 1] = np.inf; init[11, 0] = np.nan; init[15, 2] = 1e300
    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.3, 3.0, d))
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=0, n_keep_draws=4, step_size=0.25, precond_mat=M)
    g_draws, g = mcmc_amd.mala(kg, init, st, prec=prec)
    assert mcmc_amd.last_kernel().startswith("mala_gauss_dense_m_kernel")
    s = orc.make_settings(seed=5, n_burnin=0, n_keep=4, step=0.25, W=4, hoist=1, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_MALA, orc.TargetSpec(ko, d, prec=prec, W=4), init, s)
    if kind != "dense":
        assert np.isinf(o_draws[:, :, 3]).any() and np.isinf(o_draws[:, :, 8]).any(), "the case is meant to keep +-inf in single dimensions"
    assert np.isnan(o_draws[:, :, 11]).all() and np.isfinite(o_draws[:, :, 15]).all()
    assert np.isfinite(o_draws[:, :, 0]).all() and 0 < o["n_accept"].sum() < 4 * C, "... next to healthy chains that accept and reject"
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("algo", ["mala", "hmc"])
@pytest.mark.parametrize("d,N", [(70, 33), (300, 40)])
def test_logistic_chain_driven_non_finite(algo, d, N):
    """logit_lds_kernel<., MALA | HMC>: an infinite coefficient makes eta +-inf / NaN and one gradient entry infinite; the
    reference's `precond_matrix * grad_obj` / `inv_precond_matrix * mntm` then poison every coefficient."""
    C = 40
    X, y = synth.logistic_problem(d, N, seed=7)
    init = synth.initial_states(C, d, seed=13) * 0.3
    init[0, 3] = np.inf
    init[33, d - 1] = -np.inf
    init[9] *= 1.0e200
    dq = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
    st = mcmc_amd.default_settings(rng_seed_value=17, n_burnin_draws=1, n_keep_draws=3, n_leap_steps=2, step_size=0.05)
    g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=dq, eta_chains=2)
    s = orc.make_settings(seed=17, n_burnin=1, n_keep=3, n_leap=2, step=0.05, W=4, hoist=1, blocks=4, block_size=dq)
    o_draws, o = orc.run_many(orc.ALGO_MALA if algo == "mala" else orc.ALGO_HMC, t, init, s)
    assert len(_poisoned_chains(o_draws)) >= 2
    _same(g_draws, g, o_draws, o)


@pytest.mark.parametrize("hint", ["memo", "async"])
@pytest.mark.parametrize("adapt", [0, 6])
def test_plain_nuts_d128_chain_driven_non_finite(adapt, hint):
    """nuts_gauss_memo_kernel (retires a flagged chain at once and hands its lane the next one): flagged chains are replayed by the general variant (identity tables), which forms the reference's dense
    `inv_precond_matrix * mntm` (nuts.cpp:139-154); healthy chains of the same wave keep what the plain kernel wrote."""
    d, C = 128, (40 if hint == "async" else 200)
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=19)
    init[2] *= 1.0e300
    init[17, 3] = np.inf
    init[33, 100] = np.nan
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=6, n_keep_draws=5, n_adapt_draws=adapt, max_tree_depth=6, step_size=0.1)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec,
                               kernel_hint={"memo": mcmc_amd.KERNEL_AUTO, "async": mcmc_amd.KERNEL_NUTS_TICK_LOCAL}[hint])
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_%s_kernel<" % hint)
    s = orc.make_settings(seed=6, n_burnin=6, n_keep=5, n_adapt=adapt, max_depth=6, step=0.1, W=4)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    assert len(_poisoned_chains(o_draws)) >= 2
    _same(g_draws, g, o_draws, o)
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True)


@pytest.mark.parametrize("kern", ["memo"])
@pytest.mark.parametrize("d,adapt", [(128, 6), (128, 0), (48, 4), (12, 3)])
def test_nuts_with_a_diagonal_precond_mat_alone_finite_and_non_finite(d, adapt, kern):
    """nuts_gauss_memo_kernel<., true>: the plain-case kernel with two mass tables; chains that
    leave the finite regime are replayed by the general variant with the same tables (ref: src/nuts.cpp:139-154,168,202,204 with the diagonal matrices)."""
    C = 40
    prec = synth.dense_gaussian_precision(d)
    M = np.diag(np.random.default_rng(8).uniform(0.4, 2.5, d))
    init = synth.initial_states(C, d, seed=19)
    init[2] *= 1.0e300
    init[17, 3] = np.inf
    init[33, d - 5] = np.nan
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=6, n_keep_draws=5, n_adapt_draws=adapt, max_tree_depth=6, step_size=0.1,
                                   precond_mat=M)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec,
                               kernel_hint=mcmc_amd.KERNEL_AUTO)
    name = mcmc_amd.last_kernel()                         # nuts_gauss_memo_kernel<NT, DIAGM, PRE>
    assert name.startswith("nuts_gauss_%s_kernel<" % kern) and name.split("<")[1].split(",")[1].strip() == "true", name
    s = orc.make_settings(seed=6, n_burnin=6, n_keep=5, n_adapt=adapt, max_depth=6, step=0.1, W=4, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_NUTS, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    assert len(_poisoned_chains(o_draws)) >= 2 and o["n_accept"].sum() > 0
    _same(g_draws, g, o_draws, o)
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g["eps"], o["eps"], equal_nan=True)


def test_device_resident_run_replays_without_a_host_round_trip():
    """MI_MEM_DEVICE: flags, replay workspace and the literal launch ride the caller's stream"""
    import torch
    d, C = 128, 48
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=1)
    init[10] *= 1.0e300
    st = mcmc_amd.default_settings(rng_seed_value=2, n_burnin_draws=1, n_keep_draws=3, n_leap_steps=2, step_size=0.1)
    draws, info = mcmc_amd.sample_device("hmc", mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    torch.cuda.synchronize()
    s = orc.make_settings(seed=2, n_burnin=1, n_keep=3, n_leap=2, step=0.1, W=4)
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4), init, s)
    _same(draws.cpu().numpy(), dict(n_accept=info["n_accept"].cpu().numpy().astype(np.uint64)), o_draws, o)


# ---- round 6: the gradient of an ISO / DIAG target in the kernels that follow a chain through the non-finite regime themselves (the general variants: bounds, a
# diagonal or dense precond_mat -- nothing replays their chains).  The reference's target function multiplies element-wise (oracle: orc_target_kernel), so a +-inf
# coordinate stays in its own dimension; the mat-vec over the expanded diagonal put 0 * inf = NaN into every other one (hmc_dense.hpp: target_times).
def test_fuzz_case_hmc_diag_target_dense_precond_mat_chain_started_at_inf():
    """the case tests/fuzz_parity.py 300 8812 found (tests/golden/fuzz_r6_hmc_diag_target_dense_m_d64.npz: its inputs and the oracle's draws): hmc, DIAG target,
    d = 64, a dense precond_mat, eps = 1e160, one leapfrog; chain 3 starts at inf in one coordinate and ACCEPTS its proposal: +-inf in every dimension, not NaN"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_r6_hmc_diag_target_dense_m_d64.npz"))
    init, prec, M = z["init"], z["prec"], z["M"]
    d = init.shape[1]
    st = mcmc_amd.default_settings(rng_seed_value=int(z["rseed"]), n_burnin_draws=int(z["burn"]), n_keep_draws=int(z["keep"]), n_leap_steps=int(z["L"]),
                                   step_size=float(z["eps"]), precond_mat=M)
    g_draws, g = mcmc_amd.sample("hmc", mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec, chain0=int(z["chain0"]))
    assert mcmc_amd.last_kernel().startswith("hmc_gauss_mfma_kernel<4, 4, true, true")
    s = orc.make_settings(seed=int(z["rseed"]), n_burnin=int(z["burn"]), n_keep=int(z["keep"]), n_leap=int(z["L"]), step=float(z["eps"]), W=4, hoist=1, precond=M)
    o_draws, o = orc.run_many(orc.ALGO_HMC, orc.TargetSpec(orc.TARGET_DIAG, d, prec=prec, W=4), init, s, chain0=int(z["chain0"]))
    assert np.array_equal(o_draws, z["o_draws"], equal_nan=True) and np.array_equal(o["n_accept"], z["o_acc"])       # (the fixture is the oracle's)
    assert np.isinf(o_draws[0, :, 3]).sum() >= d - 1
    assert np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["n_accept"], o["n_accept"])


@pytest.mark.parametrize("algo", ["hmc", "nuts"])
@pytest.mark.parametrize("tgt", ["diag", "iso"])
@pytest.mark.parametrize("gen", ["dense_m", "diag_m", "bounds"])
@pytest.mark.parametrize("d", [64, 37])
def test_separable_targets_on_the_general_variants_keep_an_infinite_coordinate_in_its_dimension(algo, tgt, gen, d):
    C = 24
    rng = np.random.default_rng(d * 7 + len(gen))
    prec, kg, ko = (synth.ill_conditioned_diag(d, 20.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG) if tgt == "diag" else (None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)
    kw, okw = {}, {}
    if gen == "bounds":
        kind = rng.integers(1, 5, d)
        lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    else:
        M = np.diag(rng.uniform(0.3, 3.0, d))
        if gen == "dense_m": A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + M
        kw.update(precond_mat=M); okw.update(precond=M)
    init = synth.initial_states(C, d, seed=d) * 0.3
    for c in range(0, C, 2):                                   # every other chain starts with one infinite coordinate (unbounded dimensions only)
        free = np.arange(d) if gen != "bounds" else np.flatnonzero(~np.isfinite(lb) & ~np.isfinite(ub))
        init[c, int(rng.choice(free))] = np.inf if c % 4 == 0 else -np.inf
    n_bad = 0
    for eps in (1.0e160, 0.5):
        st = mcmc_amd.default_settings(rng_seed_value=d + 3, n_burnin_draws=0, n_keep_draws=2, n_leap_steps=1, step_size=eps, n_adapt_draws=0, max_tree_depth=2, **kw)
        s = orc.make_settings(seed=d + 3, n_burnin=0, n_keep=2, n_leap=1, step=eps, n_adapt=0, max_depth=2, W=4, hoist=1, **okw)
        g_draws, g = mcmc_amd.sample(algo, kg, init, st, prec=prec, chain0=5)
        o_draws, o = orc.run_many(orc.ALGO_HMC if algo == "hmc" else orc.ALGO_NUTS, orc.TargetSpec(ko, d, prec=prec, W=4), init, s, chain0=5)
        assert np.array_equal(g_draws, o_draws, equal_nan=True), mcmc_amd.last_kernel()
        assert np.array_equal(g["n_accept"], o["n_accept"])
        if algo == "nuts": assert np.array_equal(g["n_leap"], o["n_leap"])
        n_bad += int((~np.isfinite(o_draws)).any(axis=(0, 1)).sum())
    assert n_bad > 0
