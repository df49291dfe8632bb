"""User-defined device targets (include/mi_mcmc_target.hpp): examples/user_target.hip is compiled into its own library,
exactly as a user would, and driven through the generated C entry point `banana_run`.

CPU: the example cross-compiles for gfx950 and exports what the header promises; the target's kernel() -- one
__host__ __device__ member function -- works as the oracle's host callback.  GPU: hmc / mala / nuts / rwmh (with a dense
precond_mat, with box constraints) on many chains are bit-identical to the oracle driven by that same callback."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import mcmc_amd
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


class Banana(C.Structure):
    _fields_ = [("s2", C.c_double), ("b", C.c_double), ("c", C.c_double), ("rho", C.c_double)]


@pytest.fixture(scope="module")
def user_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("ut") / "libuser_target.so")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wall", "-Werror",
                           "-Wno-unused-function", f"-I{ROOT}/include", "-shared", f"{ROOT}/examples/user_target.hip",
                           f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", out])
    lib = C.CDLL(out)
    lib.banana_host_kernel.restype = C.c_double
    return lib


def test_example_target_library_builds_and_its_kernel_is_a_valid_host_callback(user_lib):
    assert hasattr(user_lib, "banana_run") and hasattr(user_lib, "banana_host_kernel")
    tgt = Banana(4.0, 0.3, 1.5, 0.2)
    v = np.array([0.7, -0.4, 1.1])
    g = np.zeros(3)
    val = user_lib.banana_host_kernel(v.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tgt))
    h = 1e-6
    num = np.zeros(3)
    for i in range(3):
        vp, vm = v.copy(), v.copy(); vp[i] += h; vm[i] -= h
        fp = user_lib.banana_host_kernel(vp.ctypes.data_as(C.POINTER(C.c_double)), None, C.byref(tgt))
        fm = user_lib.banana_host_kernel(vm.ctypes.data_as(C.POINTER(C.c_double)), None, C.byref(tgt))
        num[i] = (fp - fm) / (2 * h)
    assert np.isfinite(val) and np.allclose(g, num, rtol=1e-6, atol=1e-8)          # the analytic gradient is the gradient
    # the oracle samples it through the reference's callback contract
    s = orc.make_settings(seed=3, n_burnin=20, n_keep=50, n_leap=4, step=0.2, W=1)
    draws, info = orc.run_chain(orc.ALGO_HMC, None, np.array([0.1, 0.2, -0.1]), s, kernel=user_lib.banana_host_kernel, data=C.addressof(tgt), d=3)
    assert draws.shape == (50, 3) and np.isfinite(draws).all() and info["n_accept"] > 10


M3 = np.array([[1.4, 0.3, -0.2], [0.3, 0.9, 0.1], [-0.2, 0.1, 1.2]])
LB, UB = np.array([-np.inf, -6.0, -4.0]), np.array([5.0, 6.0, np.inf])
ALGOS = {"hmc": (0, orc.ALGO_HMC), "mala": (1, orc.ALGO_MALA), "nuts": (2, orc.ALGO_NUTS), "rwmh": (3, orc.ALGO_RWMH)}


@pytest.mark.gpu
@pytest.mark.parametrize("algo,step,L,precond,bounded", [("hmc", 0.15, 5, False, False), ("hmc", 0.12, 3, True, True), ("mala", 0.3, 1, True, False),
                                                        ("mala", 0.25, 1, False, True), ("nuts", 1.0, 1, False, False), ("nuts", 1.0, 1, True, True),
                                                        ("rwmh", 0.5, 1, True, True)])
def test_user_target_runs_many_chains_bit_exact_vs_the_oracle_with_the_same_callback(user_lib, algo, step, L, precond, bounded):
    d, Cn, burn, keep = 3, 70, 6, 14
    tgt = Banana(4.0, 0.3, 1.5, 0.2)
    init = np.random.default_rng(5).standard_normal((Cn, d)) * 0.5
    kw, okw = {}, {}
    if precond:
        kw["precond_mat"] = M3; okw["precond"] = M3
    if bounded:
        kw.update(vals_bound=1, lower_bounds=LB, upper_bounds=UB); okw.update(lower=LB, upper=UB)
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=step,
                                   n_adapt_draws=burn, max_tree_depth=6, **kw)
    theta = np.ascontiguousarray(init.T.copy())
    draws = np.zeros((keep, d, Cn))
    nacc = np.zeros(Cn, dtype=np.uint64)
    eps = np.zeros(Cn)
    ch = mcmc_amd.make_chains(theta, Cn, chain0=9, draws=draws, n_accept=nacc, step_size=eps)
    rc = user_lib.banana_run(ALGOS[algo][0], C.byref(tgt), C.byref(st), C.byref(ch), None)
    assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
    for c in range(Cn):
        s = orc.make_settings(seed=77, n_burnin=burn, n_keep=keep, n_leap=L, step=step, n_adapt=burn, max_depth=6, W=1, hoist=0,
                              chain_id=9 + c, **okw)
        o, info = orc.run_chain(ALGOS[algo][1], None, init[c], s, kernel=user_lib.banana_host_kernel, data=C.addressof(tgt), d=d)
        assert np.array_equal(draws[:, :, c], o), (algo, c)
        assert nacc[c] == info["n_accept"]
        if algo == "nuts":
            assert eps[c] == info["eps"]
    assert nacc.sum() > 0
    if bounded:
        assert ((draws >= LB[None, :, None]) & (draws <= UB[None, :, None])).all()


@pytest.mark.gpu
def test_rmhmc_on_a_target_without_a_tensor_is_refused(user_lib):
    tgt = Banana(4.0, 0.3, 1.5, 0.2)
    st = mcmc_amd.default_settings(n_burnin_draws=1, n_keep_draws=1)
    theta = np.zeros((3, 4))
    ch = mcmc_amd.make_chains(theta, 4)
    assert user_lib.banana_run(4, C.byref(tgt), C.byref(st), C.byref(ch), None) != 0
