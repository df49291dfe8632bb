"""A/B of two builds of the library on nuts_lds shapes, alternating in one process tree (GPU box):
python tools/nuts_lds_ab.py libA.so libB.so ... [dense512x4096 ...]   (paths relative to the repo root; each run is a child process with MI_MCMC_LIB set)"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.environ["MI_ROOT"])
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
kind, d, n_rows, Cn, nd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
theta0 = torch.from_numpy(np.ascontiguousarray((synth.initial_states(Cn, d, seed=3) * (0.1 if kind == "logistic" else 1.0)).T)).cuda()
st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, step_size=0.03, n_adapt_draws=nd // 2, max_tree_depth=10)
draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
nleap = torch.zeros(Cn, dtype=torch.int64, device="cuda")
if kind == "dense":
    tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(synth.dense_gaussian_precision(d)).cuda(), mem=mcmc_amd.MEM_DEVICE)
else:
    X, y = synth.logistic_problem(d, n_rows, seed=1)
    tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=torch.from_numpy(X).cuda(), y=torch.from_numpy(y).cuda(), mem=mcmc_amd.MEM_DEVICE)
best = 1e9
for rep in range(2):
    theta = theta0.clone()
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_leapfrogs=nleap, step_size=torch.zeros(Cn, dtype=torch.float64, device="cuda"), mem=mcmc_amd.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc_amd.run("nuts", tgt, st, ch)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{best * 1e3:.1f} ms  {float(nleap.double().sum()):.4e} leapfrogs  checksum {float(draws.double().sum()):.17g}  {mcmc_amd.last_kernel()}")
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
only = [a for a in sys.argv[1:] if not a.endswith(".so")]
SHAPES = [("logistic", 512, 1024, 32768, 8), ("dense", 256, 0, 4096, 20), ("dense", 512, 0, 4096, 10), ("dense", 256, 0, 65536, 20), ("logistic", 512, 1024, 16384, 10), ("dense", 512, 0, 32768, 10),
          ("logistic", 64, 1024, 16384, 20), ("logistic", 32, 256, 16384, 20), ("logistic", 20, 100, 65536, 20)]
for shp in SHAPES:
    if only and f"{shp[0]}{shp[1]}x{shp[3]}" not in only: continue
    for rnd in range(2):
        for lib in libs:
            env = dict(os.environ, MI_MCMC_LIB=os.path.join(ROOT, lib), MI_ROOT=ROOT)
            out = subprocess.run([sys.executable, "-c", CHILD] + [str(v) for v in shp], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if " ms " in l]
            print(shp, os.path.basename(lib), line[-1] if line else ("FAILED: " + out.stderr[-300:]), flush=True)
