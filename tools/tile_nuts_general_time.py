"""Bounded / diagonal-mass NUTS on configs[3]'s target, 16 384 chains: the built-in general variant (nuts_gauss_async_kernel<8, true>,
tick-local state) against the tile-route kernel with the built-in Gaussian as the user target (nuts_tile_kernel<GaussTile, true>:
register-carried leaf state + TileGen).  Same bits expected.  GPU box: python tools/tile_nuts_general_time.py"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth

out = "/tmp/libutt.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", f"-I{ROOT}/include", "-shared",
                       f"{ROOT}/examples/user_tile_target.hip", f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", out])
mcmc_amd.lib()
lib = C.CDLL(out)


class GaussTile(C.Structure):
    _fields_ = [("P", C.c_void_p), ("d", C.c_uint32)]


Cn, d = 16384, 128
dev = torch.device("cuda", 0)
P = synth.dense_gaussian_precision(d)
prec = torch.from_numpy(P).to(dev)
init = np.clip(synth.initial_states(Cn, d, seed=3) * 0.3, -1.0, 1.5)
theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
rng = np.random.default_rng(0)
kind = rng.integers(1, 5, d)
lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
for name, kw in (("bounds", dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub)), ("diag mass", dict(precond_mat=np.diag(np.linspace(0.5, 2.0, d)))),
                 ("bounds + diag mass", dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub, precond_mat=np.diag(np.linspace(0.5, 2.0, d))))):
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100, **kw)
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
    res = {}
    for route in ("built-in", "tile"):
        for rep in range(2):
            theta = theta0.clone()
            draws = torch.zeros((100, d, Cn), dtype=torch.float64, device=dev)
            ch = mcmc_amd.make_chains(theta, Cn, draws=draws, mem=mcmc_amd.MEM_DEVICE)
            torch.cuda.synchronize(); t0 = time.time()
            if route == "built-in":
                mcmc_amd.run("nuts", t, st, ch)
            else:
                rc = lib.gauss_tile_run(C.c_int(2), C.byref(GaussTile(prec.data_ptr(), d)), C.c_uint64(d), C.byref(st), C.byref(ch), C.c_void_p(0))
                assert rc == 0, mcmc_amd.lib().mi_mcmc_last_error().decode()
            torch.cuda.synchronize(); ms = (time.time() - t0) * 1e3
        res[route] = (ms, draws.clone(), mcmc_amd.last_kernel())
    print(f"{name}: built-in {res['built-in'][0]:.1f} ms ({res['built-in'][2]}), tile {res['tile'][0]:.1f} ms ({res['tile'][2]}), same bits: {bool(torch.equal(res['built-in'][1], res['tile'][1]))}")
