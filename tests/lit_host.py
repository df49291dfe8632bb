"""ctypes binding of tests/lit_host.hip: the HOST instantiation of the product's literal replay (mcmc_amd/csrc/literal.hpp).
Test infrastructure: built on first use with hipcc (which cross-compiles without a GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_SO = os.path.join(HERE, "_lit_host.so")
_ENGINE_INC = os.path.join(ROOT, "include", "mi_mcmc_engine")      # the engine headers that ship with the public ones
_SRC = ([os.path.join(HERE, "lit_host.hip")] + [os.path.join(ROOT, "mcmc_amd", "csrc", f) for f in ("literal.hpp", "literal_host.hpp", "host_linalg.hpp")]
        + [os.path.join(_ENGINE_INC, "det_math.hpp")])
_dp = C.POINTER(C.c_double)
_lib = None

KIND = {"iso": 0, "diag": 1, "dense": 2, "logit": 3, "callback": 4}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRC):
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-shared",
                                   f"-I{_ENGINE_INC}", "-o", _SO, _SRC[0]])
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


ALGO = {"hmc": 0, "mala": 1, "nuts": 2, "rwmh": 3, "rmhmc": 4}


def run(algo, kind, init, seed, n_burnin, n_keep, n_leap, eps, prec=None, X=None, y=None, chain0=0, draw0=0,
        lower=None, upper=None, precond=None, n_adapt=0, max_depth=10, delta=0.55, gamma=0.05, t0=10.0, kappa=0.75, step_in=None, n_fp=5,
        mass_diag=None, kernel_cb=None, kernel_data=None, tensor_cb=None, tensor_data=None, adapt_state_in=None):
    """algo 'hmc' | 'mala' | 'nuts' | 'rwmh'; init [C, d].  Returns (draws [n_keep, d, C], dict(n_accept, n_leap, theta [C, d], eps, depth))."""
    init = _f(init)
    Cn, d = init.shape
    theta = np.ascontiguousarray(init.T.copy())              # [d][C]
    draws = np.zeros((n_keep, d, Cn))
    nacc = np.zeros(Cn, dtype=np.uint64)
    nleap = np.zeros(Cn, dtype=np.uint64)
    prec, X, y, lower, upper, precond = _f(prec), _f(X), _f(y), _f(lower), _f(upper), _f(precond)
    n_rows = 0 if X is None else X.shape[0]
    u64p = C.POINTER(C.c_uint64)
    step = np.zeros(Cn) if step_in is None else np.array(step_in, dtype=np.float64, copy=True)
    depth = np.zeros((n_burnin + n_keep, Cn), dtype=np.uint32)
    mass_diag = _f(mass_diag)
    adapt = np.zeros((3, Cn)) if adapt_state_in is None else np.array(adapt_state_in, dtype=np.float64, copy=True)
    rc = lib().lit_host_run_ext(C.c_int(ALGO[algo]), C.c_int(KIND[kind]), C.c_uint32(d), C.c_uint32(n_rows),
                            _p(prec), _p(X), _p(y), C.c_uint64(Cn), C.c_uint64(chain0), _p(theta), _p(draws),
                            nacc.ctypes.data_as(u64p), nleap.ctypes.data_as(u64p), C.c_uint64(seed), C.c_uint32(n_burnin),
                            C.c_uint32(n_keep), C.c_uint32(n_leap), C.c_uint32(draw0), C.c_double(eps),
                            C.c_int(0 if lower is None else 1), _p(lower), _p(upper), _p(precond),
                            C.c_uint32(n_adapt), C.c_uint32(max_depth), C.c_double(delta), C.c_double(gamma), C.c_double(t0),
                            C.c_double(kappa), _p(step), depth.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(n_fp),
                            _p(mass_diag), C.c_void_p(kernel_cb or 0), C.c_void_p(kernel_data or 0), C.c_void_p(tensor_cb or 0),
                            C.c_void_p(tensor_data or 0), _p(adapt))
    assert rc == 0
    return draws, dict(n_accept=nacc, n_leap=nleap, theta=theta.T.copy(), eps=step, depth=depth, adapt_state=adapt)
