"""C++ drop-in header include/mcmc.hpp: builds on CPU; runs (and is checked) on the GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "hmc_plumbing")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", f"-I{ROOT}/include", f"{ROOT}/examples/hmc_plumbing.cpp",
           f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_reference_style_program_compiles_against_the_header(tmp_path):
    if not os.path.exists(mcmc_amd.LIB_PATH):
        pytest.skip("libmi_mcmc.so not built")
    exe = _build(tmp_path)
    if mcmc_amd.lib().mi_mcmc_device_count() == 0:
        out = subprocess.run([exe], capture_output=True, text=True)
        assert "callback ok=0" in out.stdout          # no GPU: fails loudly, never samples on the CPU
        assert out.returncode != 0


def test_settings_structs_keep_reference_names_and_defaults():
    """Field names / defaults of mcmc_structs.hpp:66-101,123-134,151-184 are present in our header."""
    src = open(os.path.join(ROOT, "include", "mcmc.hpp")).read()
    for frag in ["size_t n_burnin_draws = 1E03;", "size_t n_keep_draws = 1E03;", "size_t n_leap_steps = 1;",
                 "fp_t step_size = 1.0;", "Mat_t precond_mat;", "size_t n_accept_draws;",
                 "size_t n_adapt_draws = 1E03;", "fp_t target_accept_rate = 0.55;",
                 "size_t max_tree_depth = size_t(10);", "fp_t gamma_val = 0.05;", "fp_t t0_val = 10;",
                 "fp_t kappa_val = 0.75;", "size_t rng_seed_value = std::random_device{}();",
                 "bool vals_bound = false;", "ColVec_t lower_bounds;", "ColVec_t upper_bounds;",
                 "hmc_settings_t hmc_settings;", "nuts_settings_t nuts_settings;", "mala_settings_t mala_settings;",
                 "size_t n_fp_steps = 5;", "rmhmc_settings_t rmhmc_settings;"]:
        assert frag in src, frag
    for sig in ["hmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,",
                "mala(const ColVec_t& initial_vals", "nuts(const ColVec_t& initial_vals", "algo_settings_t& settings)",
                "rmhmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, tensor_fn_t tensor_fn, Mat_t& draws_out,"]:
        assert sig in src, sig


@pytest.mark.gpu
def test_example_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"callback ok=1 rows=1000 cols=3 mean=(\S+) (\S+) (\S+) acc=(\S+) grad_calls=(\d+) value_calls=(\d+)", out.stdout)
    assert m, out.stdout
    mean = np.array([float(m.group(i)) for i in (1, 2, 3)])
    assert np.abs(mean).max() < 0.15 and 0.9 < float(m.group(4)) <= 1.0     # examples/eigen/hmc_normal.cpp flow
    # the reference's callback pattern: 2 gradient calls per leapfrog, 1 value call per draw + 1 at setup
    assert int(m.group(5)) == 2 * 10 * 2000 and int(m.group(6)) == 2000 + 1
    for algo in ("hmc", "mala", "nuts"):
        mm = re.search(rf"device {algo} ok=1 rows=50 cols=1024 acc0=(\S+)", out.stdout)
        assert mm, out.stdout
        assert 0.2 < float(mm.group(1)) <= 1.0
    mm = re.search(r"device rmhmc ok=1 rows=200 cols=128 acc0=(\S+) mu0=(\S+) sigma0=(\S+)", out.stdout)
    assert mm, out.stdout
    assert 0.2 < float(mm.group(1)) <= 1.0 and 1.5 < float(mm.group(2)) < 3.2 and 1.5 < float(mm.group(3)) < 3.2
    # mcmc::mi355x::hmc_mass_adapted (not a reference mode): pooled masses land on the precisions; the per-chain form returns d x C masses
    mm = re.search(r"mass adapted \(pooled\) ok=1 cols=8192 acc0=(\S+) mass0/prec0=(\S+) massLast/precLast=(\S+)", out.stdout)
    assert mm, out.stdout
    assert float(mm.group(1)) > 0.5 and 0.6 < float(mm.group(2)) < 1.6 and 0.6 < float(mm.group(3)) < 1.6
    mm = re.search(r"mass adapted \(per chain\) ok=1 cols=8192 acc0=(\S+) masses=8192", out.stdout)
    assert mm, out.stdout
    # mcmc::rmhmc with host std::function callbacks (examples/eigen/rmhmc_normal.cpp's flow): the data are N(2, 2^2)
    mm = re.search(r"callback rmhmc ok=1 rows=200 cols=2 mean_mu=(\S+) mean_sigma=(\S+) acc=(\S+) grad_calls=(\d+) value_calls=(\d+) tensor_calls=(\d+)", out.stdout)
    assert mm, out.stdout
    assert 1.5 < float(mm.group(1)) < 3.2 and 1.5 < float(mm.group(2)) < 3.2 and 0.2 < float(mm.group(3)) <= 1.0     # (as the device run above)
    # the reference's callback pattern (src/rmhmc.cpp:199-272): per leapfrog step n_fp + 1 gradients and n_fp + 1 tensors; 1 value per
    # draw; at setup 1 tensor and 1 value
    assert int(mm.group(4)) == 300 * 1 * (5 + 1) and int(mm.group(5)) == 300 + 1 and int(mm.group(6)) == 300 * 1 * (5 + 1) + 1
    assert "refused=1" in out.stdout and "must both be host callbacks or both the device route" in out.stdout   # a reachable reason, no silent false
    mm = re.search(r"callback rwmh ok=1 rows=4000 cols=3 mean=(\S+) (\S+) (\S+) acc=(\S+) value_calls=(\d+)", out.stdout)
    assert mm, out.stdout                                       # mcmc::rwmh with the host std::function (value only)
    assert np.abs([float(mm.group(i)) for i in (1, 2, 3)]).max() < 0.3 and 0.2 < float(mm.group(4)) <= 1.0
    assert int(mm.group(5)) == 4500 + 1                         # one value call per draw + the initial point (src/rwmh.cpp:113,128)
    # mcmc::mala / mcmc::nuts with the host std::function (the reference's own example call pattern)
    mm = re.search(r"callback mala ok=1 rows=1000 cols=3 mean=(\S+) (\S+) (\S+) acc=(\S+) grad_calls=(\d+) value_calls=(\d+)", out.stdout)
    assert mm, out.stdout
    assert np.abs([float(mm.group(i)) for i in (1, 2, 3)]).max() < 0.25 and 0.3 < float(mm.group(4)) <= 1.0
    assert int(mm.group(5)) == 3 * 1500 and int(mm.group(6)) == 1500 + 1             # 3 gradient + 1 value callback per draw
    mm = re.search(r"callback nuts ok=1 rows=600 cols=3 mean=(\S+) (\S+) (\S+) acc=(\S+) grad_calls=(\d+) value_calls=(\d+)", out.stdout)
    assert mm, out.stdout
    assert np.abs([float(mm.group(i)) for i in (1, 2, 3)]).max() < 0.25 and 0.3 < float(mm.group(4)) <= 1.0


@pytest.mark.gpu
def test_host_callback_route_matches_oracle_and_fused_kernel_bitwise():
    """mcmc::hmc with a host callback (GPU drives everything but the callback) == oracle == fused kernel."""
    d = 3
    st = mcmc_amd.default_settings(rng_seed_value=1234, n_burnin_draws=50, n_keep_draws=100, n_leap_steps=10,
                                   step_size=0.2)
    tgt = orc.TargetSpec(orc.TARGET_ISO, d, W=4)
    cb = C.cast(orc.lib().orc_target_kernel, C.c_void_p)
    draws_cb, nacc_cb = mcmc_amd.hmc_callback(np.ones(d), cb, st, target_data=C.addressof(tgt.c))
    so = orc.make_settings(seed=1234, n_burnin=50, n_keep=100, n_leap=10, step=0.2, W=4, chain_id=0)
    t2 = orc.TargetSpec(orc.TARGET_ISO, d, W=4)
    o_draws, o = orc.run_chain(orc.ALGO_HMC, t2, np.ones(d), so)
    assert np.array_equal(np.asarray(draws_cb), o_draws) and nacc_cb == o["n_accept"]
    assert tgt.c.n_grad_calls == t2.c.n_grad_calls == 2 * 10 * 150
    assert tgt.c.n_value_calls == t2.c.n_value_calls == 151
    f_draws, f = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_ISO, np.ones((1, d)), st)
    assert np.array_equal(f_draws[:, :, 0], o_draws)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,d", [("iso", 3), ("dense", 9), ("diag", 17)])
def test_mala_rwmh_and_nuts_host_callback_routes_match_oracle_and_device_kernels_bitwise(kind, d):
    """mcmc::mala / mcmc::nuts with a host callback (ref: mala.hpp:66-73, nuts.hpp:65-72): the oracle's own target function is
    the callback; draws, accept counts and the callback pattern of the reference (mala: 3 gradient + 1 value per draw) agree
    with the oracle and with the fused device kernels."""
    prec = {"iso": None, "dense": synth.dense_gaussian_precision(d, seed=3), "diag": synth.ill_conditioned_diag(d, 20.0)}[kind]
    ko = {"iso": orc.TARGET_ISO, "dense": orc.TARGET_DENSE, "diag": orc.TARGET_DIAG}[kind]
    kg = {"iso": mcmc_amd.TARGET_GAUSS_ISO, "dense": mcmc_amd.TARGET_GAUSS_DENSE, "diag": mcmc_amd.TARGET_GAUSS_DIAG}[kind]
    x0 = np.linspace(-0.5, 0.8, d)
    cb = C.cast(orc.lib().orc_target_kernel, C.c_void_p)
    # mala
    st = mcmc_amd.default_settings(rng_seed_value=99, n_burnin_draws=20, n_keep_draws=60, step_size=0.3)
    tgt = orc.TargetSpec(ko, d, prec=prec, W=4)
    draws_cb, nacc_cb = mcmc_amd.mala_callback(x0, cb, st, target_data=C.addressof(tgt.c))
    t2 = orc.TargetSpec(ko, d, prec=prec, W=4)
    o_draws, o = orc.run_chain(orc.ALGO_MALA, t2, x0, orc.make_settings(seed=99, n_burnin=20, n_keep=60, step=0.3, W=4, hoist=0))
    assert np.array_equal(np.asarray(draws_cb), o_draws) and nacc_cb == o["n_accept"] and 0 < nacc_cb
    assert tgt.c.n_grad_calls == t2.c.n_grad_calls == 3 * 80 and tgt.c.n_value_calls == t2.c.n_value_calls == 81
    f_draws, f = mcmc_amd.mala(kg, x0[None, :], st, prec=prec)
    assert np.array_equal(f_draws[:, :, 0], o_draws)
    # rwmh (value callbacks only)
    st = mcmc_amd.default_settings(rng_seed_value=41, n_burnin_draws=20, n_keep_draws=60, step_size=0.25)
    tgt = orc.TargetSpec(ko, d, prec=prec, W=4)
    draws_cb, nacc_cb = mcmc_amd.rwmh_callback(x0, cb, st, target_data=C.addressof(tgt.c))
    t2 = orc.TargetSpec(ko, d, prec=prec, W=4)
    o_draws, o = orc.run_chain(orc.ALGO_RWMH, t2, x0, orc.make_settings(seed=41, n_burnin=20, n_keep=60, step=0.25, W=4))
    assert np.array_equal(np.asarray(draws_cb), o_draws) and nacc_cb == o["n_accept"] and 0 < nacc_cb
    assert tgt.c.n_grad_calls == t2.c.n_grad_calls == 0 and tgt.c.n_value_calls == t2.c.n_value_calls == 81
    f_draws, f = mcmc_amd.rwmh(kg, x0[None, :], st, prec=prec)
    assert np.array_equal(f_draws[:, :, 0], o_draws)
    # nuts, dual averaging on
    st = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=15, n_keep_draws=25, n_adapt_draws=15, max_tree_depth=6)
    tgt = orc.TargetSpec(ko, d, prec=prec, W=4)
    draws_cb, nacc_cb = mcmc_amd.nuts_callback(x0, cb, st, target_data=C.addressof(tgt.c))
    t2 = orc.TargetSpec(ko, d, prec=prec, W=4)
    o_draws, o = orc.run_chain(orc.ALGO_NUTS, t2, x0, orc.make_settings(seed=7, n_burnin=15, n_keep=25, n_adapt=15, max_depth=6, step=1.0, W=4))
    assert np.array_equal(np.asarray(draws_cb), o_draws) and nacc_cb == o["n_accept"]
    assert tgt.c.n_grad_calls == t2.c.n_grad_calls and tgt.c.n_value_calls == t2.c.n_value_calls
    f_draws, f = mcmc_amd.nuts(kg, x0[None, :], st, prec=prec)
    assert np.array_equal(f_draws[:, :, 0], o_draws)


@pytest.mark.gpu
def test_python_callable_as_target():
    d = 4
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=200, n_keep_draws=800, n_leap_steps=7, step_size=0.25)
    prec = np.array([1.0, 2.0, 4.0, 8.0])

    def logk(v, want_grad):
        return -0.5 * float(np.sum(prec * v * v)), (-prec * v if want_grad else None)

    draws, nacc = mcmc_amd.hmc_callback(np.zeros(d) + 0.5, logk, st)
    assert draws.shape == (800, d) and 0.6 < nacc / 800 <= 1.0
    assert np.abs(draws.var(0) * prec - 1).max() < 0.6


@pytest.mark.gpu
def test_reference_example_programs_run_on_the_device(tmp_path):
    """examples/normal_model.cpp: the flows of the reference's {hmc,mala,nuts,rmhmc}_normal.cpp examples with the device target.
    The posterior of (mu, sigma) concentrates at (xbar, sd) +- sd / sqrt(n)."""
    exe = str(tmp_path / "normal_model")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", f"-I{ROOT}/include", f"{ROOT}/examples/normal_model.cpp",
                           f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"data n=1000 xbar=(\S+) sd=(\S+)", out.stdout)
    xbar, sd = float(m.group(1)), float(m.group(2))
    for algo in ("hmc", "mala", "nuts", "rwmh"):
        mm = re.search(rf"{algo} ok=1 rows=2000 cols=512 mean_mu=(\S+) mean_sigma=(\S+) acc0=(\S+)", out.stdout)
        assert mm, out.stdout
        assert abs(float(mm.group(1)) - xbar) < 0.03 and abs(float(mm.group(2)) - sd) < 0.03, (algo, out.stdout)
        assert 0.1 < float(mm.group(3)) <= 1.0
    mm = re.search(r"rmhmc ok=1 rows=2000 cols=512 mean_mu=(\S+) mean_sigma=(\S+) acc0=(\S+)", out.stdout)
    assert mm, out.stdout                      # the reference's momentum sign: valid but slowly mixing
    assert abs(float(mm.group(1)) - xbar) < 0.6 and abs(float(mm.group(2)) - sd) < 0.6
