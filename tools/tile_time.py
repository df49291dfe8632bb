"""configs[1]'s shape through the user tile target (examples/user_tile_target.hip: GaussTile) next to the built-in kernel (GPU box):
python tools/tile_time.py"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
so = "/tmp/libuser_tile_target.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", f"-I{ROOT}/include", "-shared",
                       f"{ROOT}/examples/user_tile_target.hip", f"-L{ROOT}/mcmc_amd", "-lmi_mcmc", f"-Wl,-rpath,{ROOT}/mcmc_amd", "-o", so])
lib = C.CDLL(so)
class GaussTile(C.Structure): _fields_ = [("P", C.c_void_p), ("d", C.c_uint32)]
d, Cn = 128, 65536
P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T)).cuda()
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_leap_steps=16, step_size=0.05)
for which in ("builtin", "tile", "builtin", "tile"):
    theta = theta0.clone()
    draws = torch.empty((100, d, Cn), dtype=torch.float64, device="cuda")
    nacc = torch.zeros(Cn, dtype=torch.int64, device="cuda")
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if which == "builtin":
        mcmc_amd.run("hmc", mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE), st, ch)
    else:
        assert lib.gauss_tile_run(C.c_int(0), C.byref(GaussTile(P.data_ptr(), d)), C.c_uint64(d), C.byref(st), C.byref(ch), C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    print(f"{which}: {(time.perf_counter() - t0) * 1e3:.2f} ms  checksum {float(draws[-1].sum()):.6f} accept {float(nacc.double().mean()) / 100:.4f}")
    del draws
