"""mcmc_amd -- host-side mirror of the C ABI in include/mi_mcmc.h (libmi_mcmc.so).

The product is the gfx950 shared library; this module only binds it (ctypes) so that tests,
bench.py and Python callers can reach `mi_mcmc_{hmc,mala,nuts}_run` with numpy / torch buffers.
Names follow the reference: settings fields are those of mcmc::algo_settings_t
(/root/reference/include/misc/mcmc_structs.hpp:66-101,123-134,151-184), the entry points those of
mcmc::hmc / mcmc::mala / mcmc::nuts (/root/reference/include/mcmc/{hmc,mala,nuts}.hpp).

There is no CPU fallback: if the library is missing, or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_mcmc.so")

MI_OK, MI_ERR_BAD_ARG, MI_ERR_HIP, MI_ERR_UNSUPPORTED, MI_ERR_OOM, MI_ERR_NO_DEVICE = range(6)
TARGET_GAUSS_ISO, TARGET_GAUSS_DIAG, TARGET_GAUSS_DENSE, TARGET_LOGISTIC, TARGET_NORMAL_MODEL = 1, 2, 3, 4, 5
MEM_HOST, MEM_DEVICE = 0, 1
KERNEL_AUTO, KERNEL_ELEMENTWISE_1LANE, KERNEL_ELEMENTWISE_4LANE, KERNEL_NUTS_LOCKSTEP = 0, 1, 2, 3   # mi_kernel_hint
KERNEL_HMC_TWO_WAVES_PER_SIMD, KERNEL_HMC_ONE_WAVE_PER_SIMD, KERNEL_HMC_SPLIT2, KERNEL_HMC_SPLIT4, KERNEL_HMC_SPLIT4_TWO_WAVES = 4, 5, 6, 7, 8
KERNEL_NUTS_TICK_LOCAL, KERNEL_NUTS_REG, KERNEL_NUTS_SPLIT, KERNEL_LITERAL, KERNEL_NUTS_DYN, KERNEL_NUTS_MEMO, KERNEL_NUTS_MEMO_INTICK = 9, 10, 11, 12, 13, 14, 15

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)


class mi_target(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("d", C.c_uint64),
                ("prec", C.c_void_p), ("X", C.c_void_p), ("y", C.c_void_p),
                ("n_rows", C.c_uint64), ("mem", C.c_int32), ("kernel_hint", C.c_int32)]


class mi_settings(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("vals_bound", C.c_int32),
                ("rng_seed_value", C.c_uint64), ("lower_bounds", C.c_void_p),
                ("upper_bounds", C.c_void_p), ("n_burnin_draws", C.c_uint64),
                ("n_keep_draws", C.c_uint64), ("n_leap_steps", C.c_uint64),
                ("step_size", C.c_double), ("precond_mat", C.c_void_p),
                ("n_adapt_draws", C.c_uint64), ("target_accept_rate", C.c_double),
                ("max_tree_depth", C.c_uint64), ("gamma_val", C.c_double),
                ("t0_val", C.c_double), ("kappa_val", C.c_double), ("n_fp_steps", C.c_uint64)]


class mi_chains(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mem", C.c_int32), ("n_chains", C.c_uint64),
                ("chain0", C.c_uint64), ("theta", C.c_void_p), ("draws", C.c_void_p),
                ("n_accept", C.c_void_p), ("step_size", C.c_void_p), ("n_leapfrogs", C.c_void_p),
                ("nuts_depth", C.c_void_p), ("draw0", C.c_uint64), ("nuts_adapt_state", C.c_void_p), ("mass_diag", C.c_void_p),
                ("n_leapfrogs_executed", C.c_void_p)]


class MiMcmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mi_mcmc status {code}: {msg}")
        self.code = code


_lib = None

EXPORTS = [
    "mi_settings_default", "mi_mcmc_last_error", "mi_mcmc_last_kernel", "mi_mcmc_version", "mi_mcmc_device_count", "mi_mcmc_release_workspace", "mi_mcmc_run_user_target", "mi_mcmc_run_user_target_v", "mi_mcmc_run_tile_target",
    "mi_mcmc_hmc_run", "mi_mcmc_mala_run", "mi_mcmc_nuts_run", "mi_mcmc_rwmh_run", "mi_mcmc_rmhmc_run", "mi_mcmc_hmc_run_mass_adapted", "mi_mcmc_hmc_run_mass_adapted_per_chain", "mi_mcmc_hmc_run_callback", "mi_mcmc_mala_run_callback", "mi_mcmc_nuts_run_callback", "mi_mcmc_rwmh_run_callback", "mi_mcmc_rmhmc_run_callback",
    "mi_mcmc_draws_to_chain_major", "mi_mcmc_draws_to_chain_major_device", "mi_mcmc_shard_bounds", "mi_mcmc_allgather_draws", "mi_mcmc_allgather_draws_ragged", "mi_mcmc_merge_shards", "mi_mcmc_draw_stats",
    "mi_mcmc_allgather_draws_rank_major", "mi_mcmc_rank_major_index", "mi_mcmc_allgather_draws_begin", "mi_mcmc_allgather_draws_wait",
    "mi_mcmc_mat_inverse", "mi_mcmc_mat_cholesky_lower",
]
# test / measurement infrastructure: libmi_mcmc_probes.so (mcmc_amd/csrc/mi_mcmc_probes.h), not part of the shipped library
PROBE_EXPORTS = ["mi_probe_mfma_f64", "mi_probe_math", "mi_probe_normals", "mi_probe_uniform", "mi_probe_fp64_peak", "mi_probe_mfma_cycles"]


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and dlopen them BY PATH on `import torch`.  If the
    engine was loaded first (bound to /opt/rocm's copies) the process would then hold two HIP runtimes, and torch's streams and
    device pointers mean nothing to the second one.  So: when torch is installed but not imported yet, load its runtime
    libraries first; the engine's DT_NEEDED entries (same sonames) then resolve to them, whichever import comes first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load libmi_mcmc.so (built in-tree by __graft_entry__.build() / mcmc_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        path = os.environ.get("MI_MCMC_LIB", LIB_PATH)      # A/B builds of the same engine (tools/), never a fallback
        if not os.path.exists(path):
            raise MiMcmcError(-1, f"{path} not built: run `make -C mcmc_amd/csrc` (no CPU fallback)")
        _one_hip_runtime()
        _lib = C.CDLL(path)
        _lib.mi_mcmc_last_error.restype = C.c_char_p
        _lib.mi_mcmc_last_kernel.restype = C.c_char_p
    return _lib


_plib = None


def probes_lib():
    """Load libmi_mcmc_probes.so: the diagnostics of the GPU tests and tools/ (linked against the engine, loaded after it)."""
    global _plib
    if _plib is None:
        lib()
        path = os.path.join(os.path.dirname(LIB_PATH), "libmi_mcmc_probes.so")
        if not os.path.exists(path):
            raise MiMcmcError(-1, f"{path} not built: run `make -C mcmc_amd/csrc`")
        _plib = C.CDLL(path)
    return _plib


def _check(rc):
    if rc != MI_OK:
        raise MiMcmcError(rc, lib().mi_mcmc_last_error().decode())


def _ptr(a):
    """Address of a numpy array (host), a torch tensor (host or device), an int, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


def default_settings(**kw):
    """algo_settings_t with the reference defaults, then keyword overrides."""
    s = mi_settings()
    lib().mi_settings_default(C.byref(s))
    keep = []
    for k, v in kw.items():
        if k in ("lower_bounds", "upper_bounds", "precond_mat"):
            if v is not None:
                v = np.ascontiguousarray(v, dtype=np.float64)
                keep.append(v)
                v = v.ctypes.data
        setattr(s, k, v)
    s._keep = keep
    return s


def make_target(kind, d, prec=None, X=None, y=None, mem=MEM_HOST, kernel_hint=KERNEL_AUTO):
    t = mi_target()
    t.struct_size = C.sizeof(mi_target)
    t.kind, t.d, t.mem, t.kernel_hint = kind, int(d), mem, int(kernel_hint)
    keep = []
    if mem == MEM_HOST:
        prec = None if prec is None else np.ascontiguousarray(prec, dtype=np.float64)
        X = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
        y = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
    keep += [prec, X, y]
    t.prec, t.X, t.y = _ptr(prec), _ptr(X), _ptr(y)
    t.n_rows = int(X.shape[0]) if X is not None else (int(y.shape[0]) if (y is not None and kind == TARGET_NORMAL_MODEL) else 0)
    t._keep = keep
    return t


def make_chains(theta, n_chains, chain0=0, draws=None, n_accept=None, step_size=None, n_leapfrogs=None,
                nuts_depth=None, mem=MEM_HOST, draw0=0, mass_diag=None, nuts_adapt_state=None, n_leapfrogs_executed=None):
    c = mi_chains()
    c.struct_size = C.sizeof(mi_chains)
    c.mem, c.n_chains, c.chain0, c.draw0 = mem, int(n_chains), int(chain0), int(draw0)
    c.theta, c.draws, c.n_accept = _ptr(theta), _ptr(draws), _ptr(n_accept)
    c.step_size, c.n_leapfrogs, c.nuts_depth = _ptr(step_size), _ptr(n_leapfrogs), _ptr(nuts_depth)
    c.mass_diag = _ptr(mass_diag)                         # hmc only: per-chain diagonal masses [d][C]
    c.nuts_adapt_state = _ptr(nuts_adapt_state)           # nuts: dual-averaging state [3][C], in (continuation inside the window) / out
    c.n_leapfrogs_executed = _ptr(n_leapfrogs_executed)   # out [C]: leapfrogs really computed (nuts on the memoised kernel: fewer than n_leapfrogs)
    c._keep = [theta, draws, n_accept, step_size, n_leapfrogs, nuts_depth, mass_diag, nuts_adapt_state, n_leapfrogs_executed]
    return c


_RUN = {"hmc": "mi_mcmc_hmc_run", "mala": "mi_mcmc_mala_run", "nuts": "mi_mcmc_nuts_run", "rwmh": "mi_mcmc_rwmh_run",
        "rmhmc": "mi_mcmc_rmhmc_run"}


def hmc_mass_adapted(target, settings, chains, n_windows=3, stream=None):
    """mi_mcmc_hmc_run_mass_adapted (NOT a reference mode): hmc with a diagonal mass matrix pooled over the chains and
    re-estimated after each of n_windows parts of the burn-in.  Returns the final mass diagonal [d]."""
    mass = np.zeros(int(target.d))
    _check(lib().mi_mcmc_hmc_run_mass_adapted(C.byref(target), C.byref(settings), C.byref(chains), C.c_uint32(n_windows),
                                              C.c_void_p(mass.ctypes.data), C.c_void_p(stream or 0)))
    return mass


def hmc_mass_adapted_per_chain(target, settings, chains, n_windows=3, mass_out=None, first_step_size=0.0, stream=None):
    """mi_mcmc_hmc_run_mass_adapted_per_chain (NOT a reference mode): every chain adapts its own diagonal mass from its own burn-in
    draws.  mass_out: [d][C] in the memory space of `chains` (numpy array or torch tensor) or None; first_step_size: the step of
    part 0 (M = I on the raw target; 0 = settings.step_size, which is in the preconditioned metric)."""
    _check(lib().mi_mcmc_hmc_run_mass_adapted_per_chain(C.byref(target), C.byref(settings), C.byref(chains), C.c_uint32(n_windows),
                                                        C.c_double(first_step_size), C.c_void_p(_ptr(mass_out)), C.c_void_p(stream or 0)))
    return mass_out


def rmhmc_callback(initial_vals, kernel_fn, kernel_data, tensor_fn, tensor_data, settings):
    """mi_mcmc_rmhmc_run_callback: mcmc::rmhmc for one chain with HOST callbacks.  kernel_fn / tensor_fn: C function pointers (ctypes
    function objects or raw addresses) with the signatures of mi_log_kernel_cb / mi_tensor_cb; *_data: addresses handed through.
    Returns (draws [n_keep, d], n_accept)."""
    x0 = np.ascontiguousarray(initial_vals, dtype=np.float64)
    d, n_keep = int(x0.shape[0]), int(settings.n_keep_draws)
    out = np.zeros((d, n_keep))                           # column-major n_keep x d
    nacc = C.c_uint64(0)
    def addr(f):
        return f if isinstance(f, int) or f is None else C.cast(f, C.c_void_p).value
    _check(lib().mi_mcmc_rmhmc_run_callback(C.c_void_p(x0.ctypes.data), C.c_uint64(d), C.c_void_p(addr(kernel_fn)), C.c_void_p(addr(kernel_data)),
                                            C.c_void_p(addr(tensor_fn)), C.c_void_p(addr(tensor_data)), C.byref(settings),
                                            C.c_void_p(out.ctypes.data), C.byref(nacc)))
    return out.T.copy(), int(nacc.value)


def test_set_grid_cap(max_workgroups):
    """mi_mcmc_test_set_grid_cap (TEST HOOK, mi_mcmc_probes.h): cap the persistent grids of the dynamic-hand-out NUTS kernels; 0 = none."""
    lib().mi_mcmc_test_set_grid_cap(C.c_uint32(int(max_workgroups)))


def last_kernel():
    """mi_mcmc_last_kernel: the kernel this thread's last run spent its time in, as rocprofv3 names it."""
    return lib().mi_mcmc_last_kernel().decode()


def release_workspace(stream=None, all_streams=True):
    """mi_mcmc_release_workspace: free the cached kernel workspaces of the current device; returns the bytes freed."""
    n = C.c_uint64(0)
    _check(lib().mi_mcmc_release_workspace(C.c_void_p(stream or 0), C.c_int(1 if all_streams else 0), C.byref(n)))
    return int(n.value)


def run(algo, target, settings, chains, stream=None):
    """Raw call: mi_mcmc_<algo>_run(target, settings, chains, stream)."""
    fn = getattr(lib(), _RUN[algo])
    _check(fn(C.byref(target), C.byref(settings), C.byref(chains), C.c_void_p(stream or 0)))


def sample(algo, kind, init, settings, prec=None, X=None, y=None, chain0=0, want_draws=True, draw0=0, step_size_in=None,
           kernel_hint=KERNEL_AUTO, adapt_state_in=None, want_adapt_state=False):
    """Host-buffer convenience: init is [C, d] (row per chain, like C calls of mcmc::<algo> with
    initial_vals = init[c]).  Returns draws [n_keep, d, C] and a dict of per-chain outputs."""
    init = np.ascontiguousarray(init, dtype=np.float64)
    n_chains, d = init.shape
    theta = np.array(init.T, dtype=np.float64, order="C", copy=True)   # [d][C]; always a copy (the run overwrites it)
    n_keep = int(settings.n_keep_draws)
    draws = np.zeros((n_keep, d, n_chains)) if want_draws else None
    n_accept = np.zeros(n_chains, dtype=np.uint64)
    n_leap = np.zeros(n_chains, dtype=np.uint64)
    eps = np.zeros(n_chains) if step_size_in is None else np.array(step_size_in, dtype=np.float64, copy=True)
    n_tot = int(settings.n_burnin_draws) + n_keep
    depth = np.zeros((n_tot, n_chains), dtype=np.uint32) if algo == "nuts" else None
    adapt = None
    if algo == "nuts" and (want_adapt_state or adapt_state_in is not None):    # the dual-averaging state [3][C]: in (continuation inside the window) / out
        adapt = np.zeros((3, n_chains)) if adapt_state_in is None else np.array(adapt_state_in, dtype=np.float64, copy=True)
    t = make_target(kind, d, prec=prec, X=X, y=y, kernel_hint=kernel_hint)
    n_exec = np.zeros(n_chains, dtype=np.uint64)
    c = make_chains(theta, n_chains, chain0=chain0, draws=draws, n_accept=n_accept,
                    step_size=eps, n_leapfrogs=n_leap, nuts_depth=depth, draw0=draw0, nuts_adapt_state=adapt, n_leapfrogs_executed=n_exec)
    run(algo, t, settings, c)
    return draws, dict(n_accept=n_accept, n_leap=n_leap, eps=eps, theta=theta, depth=depth, adapt_state=adapt, n_exec=n_exec)


def sample_device(algo, kind, init, settings, prec=None, X=None, y=None, chain0=0, want_draws=True, draw0=0,
                  step_size_in=None, kernel_hint=KERNEL_AUTO, device=None, stream=None):
    """Device-resident form of sample(): chain state, draws and counters are torch tensors in HBM on `device` (default:
    the current CUDA device) and stay there -- what mcmc_amd.dist all-gathers over RCCL without a host round trip.
    init: [C, d] numpy array or torch tensor.  Returns (draws [n_keep, d, C] or None, dict of per-chain tensors)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        init_t = torch.as_tensor(init, dtype=torch.float64).to(dev)
        n_chains, d = init_t.shape
        theta = init_t.t().contiguous()                                    # [d][C]; a copy (the run overwrites it)
        if theta.data_ptr() == init_t.data_ptr():
            theta = theta.clone()
        n_keep = int(settings.n_keep_draws)
        n_tot = int(settings.n_burnin_draws) + n_keep
        draws = torch.empty((n_keep, d, n_chains), dtype=torch.float64, device=dev) if want_draws else None
        n_accept = torch.zeros(n_chains, dtype=torch.int64, device=dev)
        n_leap = torch.zeros(n_chains, dtype=torch.int64, device=dev)
        eps = (torch.zeros(n_chains, dtype=torch.float64, device=dev) if step_size_in is None
               else torch.as_tensor(step_size_in, dtype=torch.float64).to(dev).clone())
        depth = torch.zeros((n_tot, n_chains), dtype=torch.int32, device=dev) if algo == "nuts" else None
        to_dev = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float64).to(dev).contiguous()
        t = make_target(kind, d, prec=to_dev(prec), X=to_dev(X), y=to_dev(y), mem=MEM_DEVICE, kernel_hint=kernel_hint)
        c = make_chains(theta, n_chains, chain0=chain0, draws=draws, n_accept=n_accept, step_size=eps,
                        n_leapfrogs=n_leap, nuts_depth=depth, mem=MEM_DEVICE, draw0=draw0)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        run(algo, t, settings, c, stream=st)
    return draws, dict(n_accept=n_accept, n_leap=n_leap, eps=eps, theta=theta, depth=depth)


# mcmc::hmc / mcmc::mala / mcmc::nuts, many chains at once
def hmc(kind, init, settings, **kw):
    return sample("hmc", kind, init, settings, **kw)


def mala(kind, init, settings, **kw):
    return sample("mala", kind, init, settings, **kw)


def nuts(kind, init, settings, **kw):
    return sample("nuts", kind, init, settings, **kw)


def rwmh(kind, init, settings, **kw):
    """mcmc::rwmh: settings.step_size carries par_scale, settings.precond_mat carries cov_mat."""
    return sample("rwmh", kind, init, settings, **kw)


def rmhmc(kind, init, settings, **kw):
    """mcmc::rmhmc with the metric tensor built into the target kind (TARGET_NORMAL_MODEL: Fisher information)."""
    return sample("rmhmc", kind, init, settings, **kw)


LOG_KERNEL_CB = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def hmc_callback(initial_vals, callback, settings, target_data=None, algo="hmc"):
    """mcmc::hmc / mala / nuts with a host callback for one chain (mi_mcmc_<algo>_run_callback).

    callback: either a ctypes function pointer with the mi_log_kernel_cb signature or a Python
    callable f(vals: np.ndarray, want_grad: bool) -> (value, grad or None)."""
    x0 = np.ascontiguousarray(initial_vals, dtype=np.float64)
    d = x0.size
    n_keep = int(settings.n_keep_draws)
    draws = np.zeros((n_keep, d), order="F")
    n_acc = C.c_uint64(0)
    if callable(callback) and not isinstance(callback, C._CFuncPtr):
        def _tramp(vals, grad, _user):
            v = np.ctypeslib.as_array(vals, shape=(d,))
            val, g = callback(v.copy(), bool(grad))
            if grad:
                np.ctypeslib.as_array(grad, shape=(d,))[:] = g
            return float(val)
        cb = LOG_KERNEL_CB(_tramp)
    else:
        cb = callback
    fn = getattr(lib(), f"mi_mcmc_{algo}_run_callback")
    args = [C.c_void_p(x0.ctypes.data), C.c_uint64(d), C.cast(cb, C.c_void_p), C.c_void_p(target_data or 0),
            C.byref(settings), C.c_void_p(draws.ctypes.data), C.byref(n_acc)]
    if algo == "nuts":
        args.append(C.c_void_p(0))
    _check(fn(*args))
    return draws, int(n_acc.value)


def mala_callback(initial_vals, callback, settings, target_data=None):
    return hmc_callback(initial_vals, callback, settings, target_data, algo="mala")


def nuts_callback(initial_vals, callback, settings, target_data=None):
    return hmc_callback(initial_vals, callback, settings, target_data, algo="nuts")


def rwmh_callback(initial_vals, callback, settings, target_data=None):
    """settings.step_size carries par_scale; the callback is asked for the value only"""
    return hmc_callback(initial_vals, callback, settings, target_data, algo="rwmh")


# ---------------------------------------------------------------- diagnostics (GPU tests)
def probe_mfma(A, B, Cin):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    Cin = np.ascontiguousarray(Cin, dtype=np.float64)
    D = np.zeros((16, 16))
    _check(probes_lib().mi_probe_mfma_f64(C.c_void_p(A.ctypes.data), C.c_void_p(B.ctypes.data),
                                   C.c_void_p(Cin.ctypes.data), C.c_void_p(D.ctypes.data)))
    return D


def probe_math(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o1, o2 = np.empty_like(x), np.empty_like(x)
    _check(probes_lib().mi_probe_math(fn, C.c_void_p(x.ctypes.data), C.c_uint64(x.size),
                               C.c_void_p(o1.ctypes.data), C.c_void_p(o2.ctypes.data)))
    return o1, o2


def probe_normals(seed, chain, draw, stream, d):
    out = np.zeros(d)
    _check(probes_lib().mi_probe_normals(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(draw),
                                  C.c_uint32(stream), C.c_uint64(d), C.c_void_p(out.ctypes.data)))
    return out


def probe_uniform(seed, chain, draw, slot):
    out = np.zeros(1)
    _check(probes_lib().mi_probe_uniform(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(draw),
                                  C.c_uint32(slot), C.c_void_p(out.ctypes.data)))
    return float(out[0])


def probe_fp64_peak(use_mfma, iters=20000):
    out = C.c_double(0.0)
    _check(probes_lib().mi_probe_fp64_peak(int(use_mfma), int(iters), C.byref(out)))
    return out.value


def probe_mfma_cycles(waves_per_simd, use_lds, iters=20000):
    cyc, tf = C.c_double(0.0), C.c_double(0.0)
    _check(probes_lib().mi_probe_mfma_cycles(int(waves_per_simd), int(use_lds), int(iters), C.byref(cyc), C.byref(tf)))
    return cyc.value, tf.value


def mat_inverse(A):
    """INV of a d x d matrix as the engine computes it for a dense precond_mat (mi_mcmc_mat_inverse: the oracle's operation order; d >= 64 on the device)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    d = A.shape[0]
    out = np.empty((d, d))
    _check(lib().mi_mcmc_mat_inverse(C.c_void_p(A.ctypes.data), C.c_uint64(d), C.c_void_p(out.ctypes.data)))
    return out


def mat_cholesky_lower(A):
    """CHOL_LOWER likewise (mi_mcmc_mat_cholesky_lower)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    d = A.shape[0]
    out = np.empty((d, d))
    _check(lib().mi_mcmc_mat_cholesky_lower(C.c_void_p(A.ctypes.data), C.c_uint64(d), C.c_void_p(out.ctypes.data)))
    return out


def draw_stats(draws, n_keep=None, d=None, n_chains=None, mem=MEM_HOST, stream=None, want_acov=True):
    """mi_mcmc_draw_stats: pooled mean [d], autocovariance [n_keep, d], R-hat [d] and per-chain ESS [d] of a draws slab
    [n_keep, d, C] (numpy array, or a device tensor / pointer with mem=MEM_DEVICE and explicit shape).  want_acov=False: ESS and
    R-hat only -- as many lags as Geyer's sum needs, straight from HBM (acov is None in the result)."""
    if mem == MEM_HOST:
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        n_keep, d, n_chains = draws.shape
    mean = np.zeros(d)
    acov = np.zeros((n_keep, d)) if want_acov else None
    rhat, ess = np.zeros(d), np.zeros(d)
    _check(lib().mi_mcmc_draw_stats(C.c_void_p(_ptr(draws)), C.c_int32(mem), C.c_uint64(n_keep), C.c_uint64(d), C.c_uint64(n_chains),
                                    C.c_void_p(mean.ctypes.data), C.c_void_p(acov.ctypes.data if want_acov else 0), C.c_void_p(rhat.ctypes.data),
                                    C.c_void_p(ess.ctypes.data), C.c_void_p(stream or 0)))
    return dict(mean=mean, acov=acov, rhat=rhat, ess=ess)


def draws_to_chain_major_device(draws, n_keep, d, n_chains, out, stream=None):
    """mi_mcmc_draws_to_chain_major_device: slab [n_keep][d][C] in HBM -> [C][d][n_keep] in HBM (device tensors / pointers)."""
    _check(lib().mi_mcmc_draws_to_chain_major_device(C.c_void_p(_ptr(draws)), C.c_uint64(n_keep), C.c_uint64(d), C.c_uint64(n_chains),
                                                     C.c_void_p(_ptr(out)), C.c_void_p(stream or 0)))
