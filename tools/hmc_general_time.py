"""Timing of the general variant of hmc / mala / rwmh (argument) (diagonal precond_mat; with and without bounds) next to the plain kernel, BASELINE configs[1] shape."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
ALGO = sys.argv[1] if len(sys.argv) > 1 else "hmc"
C, d = 65536, 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
init = synth.initial_states(C, d, seed=3)
init[:, :12] = np.clip(init[:, :12], -2.9, 2.9)     # inside the bounds below: a chain that starts outside is NaN from the transform on (and replayed literally)
theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
M = np.diag(np.linspace(0.5, 2.0, d))
lb = np.full(d, -np.inf); ub = np.full(d, np.inf); lb[:8] = -3.0; ub[4:12] = 3.0
cases = {"plain": {}, "diag precond": dict(precond_mat=M), "diag precond + 12 bounded dims": dict(precond_mat=M, vals_bound=1, lower_bounds=lb, upper_bounds=ub)}
for name, kw in cases.items():
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_leap_steps=16, step_size=(0.05 if ALGO == "hmc" else 0.3), **kw)
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
    for rep in range(2):
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, C, mem=mcmc_amd.MEM_DEVICE)
        torch.cuda.synchronize(); t0 = time.time()
        mcmc_amd.run(ALGO, t, st, ch); torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
    print("%s %-32s %8.1f ms" % (ALGO, name, ms))
