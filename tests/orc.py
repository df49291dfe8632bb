"""ctypes binding of oracle/liboracle.so (CPU oracle; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("ORC_LIB", os.path.join(ROOT, "oracle", "liboracle.so"))    # bench.py's timing build: liboracle_fast.so

TARGET_ISO, TARGET_DIAG, TARGET_DENSE, TARGET_LOGISTIC, TARGET_NORMAL_MODEL = 1, 2, 3, 4, 5
ALGO_HMC, ALGO_MALA, ALGO_NUTS, ALGO_RWMH, ALGO_RMHMC = 0, 1, 2, 3, 4
ALGO_NUTS_MEMO_XD = 6                       # orc_nuts_memo_xd: ... and across the doublings of a draw (same direction, no accepted proposal in between): same bits again
ALGO_NUTS_MEMO = 5                          # orc_nuts_memo: the same run, every doubling on a memoised trajectory (same bits)   # rwmh: step = par_scale, precond = cov_mat

_dp = C.POINTER(C.c_double)


class Target(C.Structure):
    _fields_ = [("kind", C.c_int), ("d", C.c_size_t), ("prec", _dp), ("X", _dp), ("y", _dp),
                ("n_rows", C.c_size_t), ("reduce_width", C.c_int), ("reduce_blocks", C.c_int),
                ("reduce_block_size", C.c_size_t), ("eta_chains", C.c_int),
                ("n_grad_calls", C.c_uint64), ("n_value_calls", C.c_uint64), ("prec_t", _dp)]


class Settings(C.Structure):
    _fields_ = [("rng_seed_value", C.c_uint64), ("vals_bound", C.c_int),
                ("lower_bounds", _dp), ("upper_bounds", _dp),
                ("n_burnin_draws", C.c_size_t), ("n_keep_draws", C.c_size_t),
                ("n_leap_steps", C.c_size_t), ("step_size", C.c_double), ("precond_mat", _dp),
                ("n_adapt_draws", C.c_size_t), ("target_accept_rate", C.c_double),
                ("max_tree_depth", C.c_size_t), ("gamma_val", C.c_double), ("t0_val", C.c_double),
                ("kappa_val", C.c_double), ("n_fp_steps", C.c_size_t), ("reduce_width", C.c_int), ("reduce_blocks", C.c_int),
                ("reduce_block_size", C.c_size_t),
                ("hoist_factorizations", C.c_int), ("chain_id", C.c_uint64), ("work_mode", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n_accept_draws", C.c_size_t), ("n_leapfrogs", C.c_uint64),
                ("final_step_size", C.c_double), ("accept_trace", C.POINTER(C.c_uint8)),
                ("depth_trace", C.POINTER(C.c_uint32)), ("leap_trace", C.POINTER(C.c_uint32)),
                ("eps_trace", _dp)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.orc_dot.restype = C.c_double
        _lib.orc_uniform.restype = C.c_double
        _lib.orc_dmvnorm_log.restype = C.c_double
        _lib.orc_log_jacobian.restype = C.c_double
        _lib.orc_target_kernel.restype = C.c_double
        _lib.orc_mala_prop_adjustment_eval.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class TargetSpec:
    """Holds the numpy buffers alive next to the C struct."""

    def __init__(self, kind, d, prec=None, X=None, y=None, W=4, blocks=0, block_size=0, eta_chains=1):
        self.kind, self.d, self.W = kind, int(d), W
        if blocks > 1 and block_size > 0 and blocks * block_size < int(d):
            raise ValueError(f"reduction blocking {blocks} x {block_size} does not cover d = {d}: the blocked sums would drop dimensions")
        self.prec, self.X, self.y = _f64(prec), _f64(X), _f64(y)
        n_rows = self.X.shape[0] if self.X is not None else (self.y.shape[0] if self.y is not None else 0)
        self.c = Target(kind, self.d, _p(self.prec), _p(self.X), _p(self.y), n_rows, W, blocks, block_size, eta_chains, 0, 0, None)

    def kernel(self, theta, want_grad=True):
        theta = _f64(theta)
        g = np.empty(self.d) if want_grad else None
        v = lib().orc_target_kernel(_p(theta), _p(g), C.byref(self.c))
        return v, g


def make_settings(seed=1, n_burnin=0, n_keep=10, n_leap=1, step=1.0, precond=None, n_adapt=1000,
                  delta=0.55, max_depth=10, gamma=0.05, t0=10.0, kappa=0.75, W=4, hoist=1, chain_id=0,
                  lower=None, upper=None, blocks=0, block_size=0, n_fp=5, work_mode=0):
    keep = dict(precond=_f64(precond), lower=_f64(lower), upper=_f64(upper))
    s = Settings(seed, 0 if lower is None else 1, _p(keep["lower"]), _p(keep["upper"]),
                 n_burnin, n_keep, n_leap, step, _p(keep["precond"]), n_adapt, delta, max_depth,
                 gamma, t0, kappa, n_fp, W, blocks, block_size, hoist, chain_id, work_mode)
    s._keep = keep
    s._blocking = (blocks, block_size)
    return s


def run_chain(algo, target, init, settings, traces=False, kernel=None, data=None, d=None):
    """One chain through orc_hmc / orc_mala / orc_nuts. Returns (draws[n_keep,d], info dict).
    kernel / data: any callback with the reference's contract (a C function pointer + its void* target_data) instead of the
    built-in targets -- what a user-defined target library exports (tests/test_user_target.py)."""
    d = target.d if d is None else d
    init = _f64(init)
    n_keep = settings.n_keep_draws
    n_tot = settings.n_burnin_draws + n_keep
    draws = np.zeros((max(n_keep, 1), d))
    st = Stats()
    acc = np.zeros(n_tot, dtype=np.uint8)
    dep = np.zeros(n_tot, dtype=np.uint32)
    lea = np.zeros(n_tot, dtype=np.uint32)
    eps = np.zeros(n_tot)
    if traces:
        st.accept_trace = acc.ctypes.data_as(C.POINTER(C.c_uint8))
        st.depth_trace = dep.ctypes.data_as(C.POINTER(C.c_uint32))
        st.leap_trace = lea.ctypes.data_as(C.POINTER(C.c_uint32))
        st.eps_trace = _p(eps)
    kern = C.cast(lib().orc_target_kernel, C.c_void_p) if kernel is None else C.cast(kernel, C.c_void_p)
    tdata = C.byref(target.c) if data is None else C.c_void_p(data)
    if algo == ALGO_RMHMC:
        tens = C.cast(lib().orc_target_tensor, C.c_void_p)
        rc = lib().orc_rmhmc(_p(init), C.c_size_t(d), kern, tens, C.byref(target.c), C.byref(target.c),
                             C.byref(settings), _p(draws), C.byref(st))
    else:
        n_exec = C.c_uint64(0)
        if algo in (ALGO_NUTS_MEMO, ALGO_NUTS_MEMO_XD):
            fn = lib().orc_nuts_memo if algo == ALGO_NUTS_MEMO else lib().orc_nuts_memo_xd
            rc = fn(_p(init), C.c_size_t(d), kern, tdata, C.byref(settings), _p(draws), C.byref(st), C.byref(n_exec))
        else:
            fn = [lib().orc_hmc, lib().orc_mala, lib().orc_nuts, lib().orc_rwmh][algo]
            rc = fn(_p(init), C.c_size_t(d), kern, tdata, C.byref(settings), _p(draws), C.byref(st))
    assert rc == 0
    info = dict(n_accept=st.n_accept_draws, n_leap=st.n_leapfrogs, eps=st.final_step_size,
                accept=acc, depth=dep, leaps=lea, eps_trace=eps)
    if algo in (ALGO_NUTS_MEMO, ALGO_NUTS_MEMO_XD):
        info["n_exec"] = int(n_exec.value)
    return draws[:n_keep], info


def run_many(algo, target, init, settings, chain0=0, n_threads=0, want_draws=True):
    """n_chains chains; init [n_chains, d]; draws layout [n_keep, d, n_chains]."""
    init = _f64(init)
    n_chains, d = init.shape
    n_keep = settings.n_keep_draws
    draws = np.zeros((n_keep, d, n_chains)) if want_draws else None
    nacc = np.zeros(n_chains, dtype=np.uint64)
    nleap = np.zeros(n_chains, dtype=np.uint64)
    eps = np.zeros(n_chains)
    rc = lib().orc_run_many(algo, C.byref(target.c), C.byref(settings), C.c_size_t(n_chains),
                            C.c_uint64(chain0), _p(init), _p(draws),
                            nacc.ctypes.data_as(C.POINTER(C.c_uint64)),
                            nleap.ctypes.data_as(C.POINTER(C.c_uint64)), _p(eps), n_threads)
    assert rc == 0
    return draws, dict(n_accept=nacc, n_leap=nleap, eps=eps)


def math_eval(fn, x):
    x = _f64(x)
    out = np.empty_like(x)
    out2 = np.empty_like(x)
    lib().orc_math_eval(fn, _p(x), C.c_size_t(x.size), _p(out), _p(out2))
    return out, out2


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    u32 = C.POINTER(C.c_uint32)
    lib().orc_philox_eval(c.ctypes.data_as(u32), k.ctypes.data_as(u32), o.ctypes.data_as(u32))
    return o


def normal_vec(seed, chain, draw, stream, d):
    out = np.zeros(d)
    lib().orc_normal_vec(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(draw), C.c_uint32(stream),
                         C.c_size_t(d), _p(out))
    return out


def uniform(seed, chain, draw, slot):
    return lib().orc_uniform(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(draw), C.c_uint32(slot))


def dot(x, y, W):
    x, y = _f64(x), _f64(y)
    return lib().orc_dot(_p(x), _p(y), C.c_size_t(x.size), W)


def mala_prop_adjustment(target, settings, prop, prev):
    """mala_prop_adjustment(prop_vals, prev_vals) (ref: include/mcmc/mala.ipp:30-70) of a built-in target"""
    prop, prev = _f64(prop), _f64(prev)
    return lib().orc_mala_prop_adjustment_eval(C.byref(target.c), C.byref(settings), _p(prop), _p(prev))
