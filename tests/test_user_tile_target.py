"""User-defined targets on the TILED engine (include/mi_mcmc_tile_target.hpp): examples/user_tile_target.hip is compiled into its own
library, exactly as a user would, and driven through the generated C entry points.

CPU: the example cross-compiles for gfx950; its host callback is a valid target for the oracle.  GPU: (1) the dense Gaussian written
as a user tile target reproduces the built-in hmc_gauss_mfma_kernel / mala_gauss_mfma_kernel bit for bit; (2) a non-Gaussian d = 64
target (twisted Gaussian) under hmc, mala and nuts is bit-identical to the oracle driven by the same arithmetic as the reference's host
callback (ref: include/mcmc/hmc.hpp:42-48)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


class GaussTile(C.Structure):
    _fields_ = [("P", C.c_void_p), ("d", C.c_uint32)]


class TwistedTile(C.Structure):
    _fields_ = [("P", C.c_void_p), ("d", C.c_uint32), ("b", C.c_double), ("s2", C.c_double)]


@pytest.fixture(scope="module")
def tile_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("utt") / "libuser_tile_target.so")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wall", "-Werror",
                           "-Wno-unused-function", *os.environ.get("MI_TILE_CFLAGS", "").split(), f"-I{ROOT}/include", "-shared", f"{ROOT}/examples/user_tile_target.hip",
                           f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", out])
    mcmc_amd.lib()                                  # the engine first: one HIP runtime for torch, the engine and the target library
    lib = C.CDLL(out)
    lib.twisted_host_kernel.restype = C.c_double
    return lib


def _twisted_precision(d, seed=3):
    return synth.dense_gaussian_precision(d, seed=seed)


def test_example_tile_library_builds_and_its_host_callback_is_a_valid_target(tile_lib):
    for name in ("gauss_tile_run", "twisted_tile_run", "twisted_host_kernel"):
        assert hasattr(tile_lib, name)
    d = 10
    P = _twisted_precision(d)
    tgt = TwistedTile(P.ctypes.data, d, 0.2, 1.5)
    v = np.random.default_rng(0).standard_normal(d)
    g = np.zeros(d)
    dp = C.POINTER(C.c_double)
    val = tile_lib.twisted_host_kernel(v.ctypes.data_as(dp), g.ctypes.data_as(dp), C.byref(tgt))
    num, h = np.zeros(d), 1e-6
    for i in range(d):
        vp, vm = v.copy(), v.copy(); vp[i] += h; vm[i] -= h
        num[i] = (tile_lib.twisted_host_kernel(vp.ctypes.data_as(dp), None, C.byref(tgt))
                  - tile_lib.twisted_host_kernel(vm.ctypes.data_as(dp), None, C.byref(tgt))) / (2 * h)
    assert np.isfinite(val) and np.allclose(g, num, rtol=1e-5, atol=1e-7)       # the analytic gradient is the gradient
    s = orc.make_settings(seed=3, n_burnin=10, n_keep=30, n_leap=4, step=0.15, W=4)
    draws, info = orc.run_chain(orc.ALGO_HMC, None, v * 0.3, s, kernel=tile_lib.twisted_host_kernel, data=C.addressof(tgt), d=d)
    assert draws.shape == (30, d) and np.isfinite(draws).all() and info["n_accept"] > 5


def _run_tile(lib, fn, algo, target, d, init, st):
    import torch
    Cn = init.shape[0]
    theta = np.ascontiguousarray(init.T.copy())
    n_keep = int(st.n_keep_draws)
    draws = np.zeros((n_keep, d, Cn)); nacc = np.zeros(Cn, dtype=np.uint64); nleap = np.zeros(Cn, dtype=np.uint64)
    eps = np.zeros(Cn); depth = np.zeros((int(st.n_burnin_draws) + n_keep, Cn), dtype=np.uint32)
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, n_leapfrogs=nleap, step_size=eps if algo == 2 else None,
                              nuts_depth=depth if algo == 2 else None)
    rc = getattr(lib, fn)(C.c_int(algo), C.byref(target), C.c_uint64(d), C.byref(st), C.byref(ch), C.c_void_p(0))
    assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
    torch.cuda.synchronize()
    return draws, dict(n_accept=nacc, n_leap=nleap, eps=eps, depth=depth)


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("d,C_", [(128, 200), (100, 33), (70, 16)])
def test_dense_gaussian_as_a_user_tile_target_reproduces_the_built_in_kernel(tile_lib, algo, d, C_):
    import torch
    P = synth.dense_gaussian_precision(d, seed=d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=5)
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=3, n_keep_draws=6, n_leap_steps=5, step_size=0.07)
    t_draws, t = _run_tile(tile_lib, "gauss_tile_run", 0 if algo == "hmc" else 1, GaussTile(Pd.data_ptr(), d), d, init, st)
    assert mcmc_amd.last_kernel().startswith(f"{algo}_tile_kernel<")
    b_draws, b = mcmc_amd.sample(algo, mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=P)
    assert np.array_equal(t_draws, b_draws) and np.array_equal(t["n_accept"], b["n_accept"])
    assert 0 < t["n_accept"].sum()


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["hmc", "mala"])
@pytest.mark.parametrize("d,C_", [(64, 40), (37, 17)])
def test_twisted_gaussian_tile_target_against_the_oracle_with_the_same_callback(tile_lib, algo, d, C_):
    import torch
    P = _twisted_precision(d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=7) * 0.5
    eps = 0.08 if algo == "hmc" else 0.15
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=3, n_keep_draws=8, n_leap_steps=4, step_size=eps)
    g_draws, g = _run_tile(tile_lib, "twisted_tile_run", 0 if algo == "hmc" else 1, TwistedTile(Pd.data_ptr(), d, 0.2, 1.5), d, init, st)
    host = TwistedTile(P.ctypes.data, d, 0.2, 1.5)
    s = orc.make_settings(seed=4, n_burnin=3, n_keep=8, n_leap=4, step=eps, W=4, hoist=1)
    o_draws = np.zeros_like(g_draws); o_acc = np.zeros(C_, dtype=np.uint64)
    for c in range(C_):
        s.chain_id = c
        dr, info = orc.run_chain(orc.ALGO_HMC if algo == "hmc" else orc.ALGO_MALA, None, init[c], s, kernel=tile_lib.twisted_host_kernel,
                                 data=C.addressof(host), d=d)
        o_draws[:, :, c] = dr; o_acc[c] = info["n_accept"]
    assert np.array_equal(g["n_accept"], o_acc) and np.array_equal(g_draws, o_draws)
    assert 0 < o_acc.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("d,C_,max_depth", [(128, 150, 6), (100, 33, 10), (70, 16, 3)])
def test_dense_gaussian_tile_target_under_nuts_reproduces_the_built_in_nuts_kernel(tile_lib, d, C_, max_depth):
    """mcmc::nuts on the tile route (nuts_tile.hpp): with the built-in dense Gaussian as the user target, p + (e grad) / 2 is
    p - (e P theta) / 2 to the bit, so trees, step sizes and draws equal the built-in kernel's (nuts_gauss_memo_kernel, which executes fewer of the leapfrogs)."""
    import torch
    P = synth.dense_gaussian_precision(d, seed=d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=5)
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=4, n_keep_draws=5, n_adapt_draws=6, max_tree_depth=max_depth)
    t_draws, t = _run_tile(tile_lib, "gauss_tile_run", 2, GaussTile(Pd.data_ptr(), d), d, init, st)
    assert mcmc_amd.last_kernel().startswith("nuts_tile_kernel<")
    b_draws, b = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=P, kernel_hint=mcmc_amd.KERNEL_AUTO)
    bad = np.nonzero((t["depth"] != b["depth"]).any(axis=0) | (t["n_leap"] != b["n_leap"]))[0]
    assert np.array_equal(t["depth"], b["depth"]) and np.array_equal(t["n_leap"], b["n_leap"]), (bad, t["depth"][:, bad[:3]].T, b["depth"][:, bad[:3]].T)
    assert np.array_equal(t["eps"], b["eps"]) and np.array_equal(t["n_accept"], b["n_accept"])
    assert np.array_equal(t_draws, b_draws)


@pytest.mark.gpu
def test_tile_target_nuts_runs_are_cut_into_pieces_on_a_small_grid(tile_lib):
    """more chains than the chain slots of the persistent grid (capped at one workgroup here: 64 slots) and 8+ draws: the engine cuts the runs of a user target's
    nuts_tile_kernel into pieces that migrate between slots (mi_mcmc_run_tile_target: nuts_tile_setup_pieces); same results as the built-in kernel's uncut run"""
    import torch
    d, C_ = 128, 200
    P = synth.dense_gaussian_precision(d, seed=d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=5)
    st = mcmc_amd.default_settings(rng_seed_value=9, n_burnin_draws=8, n_keep_draws=9, n_adapt_draws=11, max_tree_depth=6)
    b_draws, b = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=P, kernel_hint=mcmc_amd.KERNEL_AUTO)
    mcmc_amd.test_set_grid_cap(1)
    try:
        t_draws, t = _run_tile(tile_lib, "gauss_tile_run", 2, GaussTile(Pd.data_ptr(), d), d, init, st)
    finally:
        mcmc_amd.test_set_grid_cap(0)
    assert mcmc_amd.last_kernel().startswith("nuts_tile_kernel<")
    assert np.array_equal(t["depth"], b["depth"]) and np.array_equal(t["n_leap"], b["n_leap"])
    assert np.array_equal(t["eps"], b["eps"]) and np.array_equal(t["n_accept"], b["n_accept"])
    assert np.array_equal(t_draws, b_draws)


@pytest.mark.gpu
@pytest.mark.parametrize("d,C_,adapt", [(64, 40, 6), (37, 17, 0), (2, 70, 4)])
def test_twisted_gaussian_tile_target_under_nuts_against_the_oracle_with_the_same_callback(tile_lib, d, C_, adapt):
    """a non-Gaussian d = 64 tile target under mcmc::nuts, bit for bit against the oracle driven by the same
    function as the reference's host callback (ref: include/mcmc/nuts.hpp:65-72): trees, leapfrog counts, adapted step sizes, draws."""
    import torch
    P = _twisted_precision(d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=7) * 0.5
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=4, n_keep_draws=8, n_adapt_draws=adapt, max_tree_depth=6,
                                   step_size=1.0 if adapt else 0.1)
    g_draws, g = _run_tile(tile_lib, "twisted_tile_run", 2, TwistedTile(Pd.data_ptr(), d, 0.2, 1.5), d, init, st)
    host = TwistedTile(P.ctypes.data, d, 0.2, 1.5)
    s = orc.make_settings(seed=4, n_burnin=4, n_keep=8, n_adapt=adapt, max_depth=6, step=1.0 if adapt else 0.1, W=4)
    o_draws = np.zeros_like(g_draws); o_acc = np.zeros(C_, dtype=np.uint64); o_leap = np.zeros(C_, dtype=np.uint64); o_eps = np.zeros(C_)
    o_depth = np.zeros_like(g["depth"])
    for c in range(C_):
        s.chain_id = c
        dr, info = orc.run_chain(orc.ALGO_NUTS, None, init[c], s, traces=True, kernel=tile_lib.twisted_host_kernel, data=C.addressof(host), d=d)
        o_draws[:, :, c] = dr; o_acc[c] = info["n_accept"]; o_leap[c] = info["n_leap"]; o_eps[c] = info["eps"]; o_depth[:, c] = info["depth"]
    assert np.array_equal(g["depth"], o_depth) and np.array_equal(g["n_leap"], o_leap)
    assert np.array_equal(g["eps"], o_eps) and np.array_equal(g["n_accept"], o_acc)
    assert np.array_equal(g_draws, o_draws)
    assert g["depth"].max() >= 2


def _mixed_bounds(d, seed):
    """A mix of the four bounds types of determine_bounds_type.hpp:27-57."""
    rng = np.random.default_rng(seed)
    kind = rng.integers(1, 5, d)
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf)
    ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    return lb, ub


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["hmc", "nuts"])
@pytest.mark.parametrize("d,C_,bounded,precond", [(64, 40, True, False), (37, 17, False, True), (50, 20, True, True)])
def test_twisted_tile_target_with_bounds_and_a_diagonal_precond_mat_against_the_oracle(tile_lib, algo, d, C_, bounded, precond):
    """settings.vals_bound and / or a diagonal precond_mat on the tile route (TileGen: hmc_tile_gen_kernel, nuts_tile_kernel<., true>):
    the non-Gaussian tile target against the oracle driven by the same function as the reference's host callback, bit for bit --
    draws reported in the constrained space, accept counts, and for nuts trees, leapfrog counts and adapted step sizes."""
    import torch
    P = _twisted_precision(d)
    Pd = torch.from_numpy(P).cuda()
    init = np.clip(synth.initial_states(C_, d, seed=7) * 0.3, -1.0, 1.5)       # inside every box
    kw, okw = {}, {}
    if bounded:
        lb, ub = _mixed_bounds(d, seed=d)
        kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
    if precond:
        M = np.diag(np.linspace(0.5, 2.0, d))
        kw.update(precond_mat=M); okw.update(precond=M)
    if algo == "hmc":
        st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=3, n_keep_draws=8, n_leap_steps=4, step_size=0.08, **kw)
        s = orc.make_settings(seed=4, n_burnin=3, n_keep=8, n_leap=4, step=0.08, W=4, hoist=1, **okw)
    else:
        st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=4, n_keep_draws=6, n_adapt_draws=5, max_tree_depth=5, **kw)
        s = orc.make_settings(seed=4, n_burnin=4, n_keep=6, n_adapt=5, max_depth=5, W=4, **okw)
    g_draws, g = _run_tile(tile_lib, "twisted_tile_run", 0 if algo == "hmc" else 2, TwistedTile(Pd.data_ptr(), d, 0.2, 1.5), d, init, st)
    assert mcmc_amd.last_kernel().startswith(f"{algo}_tile_gen_kernel<")
    host = TwistedTile(P.ctypes.data, d, 0.2, 1.5)
    o_draws = np.zeros_like(g_draws); o_acc = np.zeros(C_, dtype=np.uint64); o_leap = np.zeros(C_, dtype=np.uint64); o_eps = np.zeros(C_)
    for c in range(C_):
        s.chain_id = c
        dr, info = orc.run_chain(orc.ALGO_HMC if algo == "hmc" else orc.ALGO_NUTS, None, init[c], s, kernel=tile_lib.twisted_host_kernel,
                                 data=C.addressof(host), d=d)
        o_draws[:, :, c] = dr; o_acc[c] = info["n_accept"]; o_leap[c] = info["n_leap"]; o_eps[c] = info["eps"]
    assert np.array_equal(g["n_accept"], o_acc)
    if algo == "nuts":
        assert np.array_equal(g["n_leap"], o_leap) and np.array_equal(g["eps"], o_eps)
    assert np.array_equal(g_draws, o_draws)
    if bounded:
        assert ((g_draws >= lb[None, :, None]) & (g_draws <= ub[None, :, None])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("d,C_", [(64, 40), (37, 17), (50, 70)])
def test_twisted_tile_target_under_mala_with_a_diagonal_precond_mat_against_the_oracle(tile_lib, d, C_):
    """round 6: mcmc::mala with a DIAGONAL precond_mat on the tile route (mala_tile_kernel<T, true>; ref: src/mala.cpp:57-58,123,159,
    include/mcmc/mala.ipp:58-64, include/stats/dmvnorm.hpp:28-54): mu(v) = v + (eps^2 (M g)) / 2, proposal = mu + eps (sqrt(M) z), Sigma = eps^2 M in both
    dmvnorm terms -- the non-Gaussian tile target against the oracle driven by the same function as the reference's host callback, bit for bit, one
    chain started in the non-finite regime included (this route applies the NaN rules of the reference's dense products itself: no replay)."""
    import torch
    P = _twisted_precision(d)
    Pd = torch.from_numpy(P).cuda()
    init = synth.initial_states(C_, d, seed=7) * 0.5
    init[3] *= 1e200; init[5, 1] = np.inf
    M = np.diag(np.random.default_rng(d).uniform(0.4, 2.5, d))
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=6, n_keep_draws=10, step_size=0.3, precond_mat=M)
    g_draws, g = _run_tile(tile_lib, "twisted_tile_run", 1, TwistedTile(Pd.data_ptr(), d, 0.2, 1.5), d, init, st)
    assert mcmc_amd.last_kernel().startswith("mala_tile_gen_kernel<")
    host = TwistedTile(P.ctypes.data, d, 0.2, 1.5)
    s = orc.make_settings(seed=4, n_burnin=6, n_keep=10, step=0.3, W=4, hoist=1, precond=M)
    o_draws = np.zeros_like(g_draws); o_acc = np.zeros(C_, dtype=np.uint64)
    for c in range(C_):
        s.chain_id = c
        dr, info = orc.run_chain(orc.ALGO_MALA, None, init[c], s, kernel=tile_lib.twisted_host_kernel, data=C.addressof(host), d=d)
        o_draws[:, :, c] = dr; o_acc[c] = info["n_accept"]
    assert np.array_equal(g["n_accept"], o_acc) and np.array_equal(g_draws, o_draws, equal_nan=True)
    assert 0 < o_acc.sum() and (d < 64 or o_acc.sum() < 10 * C_)          # accepts and (at the larger d) rejections


@pytest.mark.gpu
def test_tile_route_refuses_what_it_does_not_implement(tile_lib):
    import torch
    d = 64
    Pd = torch.from_numpy(_twisted_precision(d)).cuda()
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1, vals_bound=1, lower_bounds=np.full(d, -1.0), upper_bounds=np.full(d, 1.0))
    theta = np.zeros((d, 4))
    ch = mcmc_amd.make_chains(theta, 4)
    rc = tile_lib.twisted_tile_run(C.c_int(1), C.byref(TwistedTile(Pd.data_ptr(), d, 0.2, 1.5)), C.c_uint64(d), C.byref(st), C.byref(ch), C.c_void_p(0))
    assert rc == mcmc_amd.MI_ERR_UNSUPPORTED                            # mala with bounds: the one-lane route
    Mfull = _twisted_precision(d)                                       # a dense precond_mat: likewise
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1, precond_mat=Mfull)
    rc = tile_lib.twisted_tile_run(C.c_int(0), C.byref(TwistedTile(Pd.data_ptr(), d, 0.2, 1.5)), C.c_uint64(d), C.byref(st), C.byref(ch), C.c_void_p(0))
    assert rc == mcmc_amd.MI_ERR_UNSUPPORTED
    for algo, kw in ((3, {}), (2, dict(max_tree_depth=11))):          # rwmh is not on this route; nuts trees deeper than its records
        rc = tile_lib.twisted_tile_run(C.c_int(algo), C.byref(TwistedTile(Pd.data_ptr(), d, 0.2, 1.5)), C.c_uint64(d),
                                       C.byref(mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1, **kw)), C.byref(ch), C.c_void_p(0))
        assert rc == mcmc_amd.MI_ERR_UNSUPPORTED
