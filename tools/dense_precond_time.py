"""Timing of the dense-precond_mat variants (hmc, mala, nuts, rwmh) on configs[1]'s shape next to the plain kernels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
d = 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
A = np.random.default_rng(5).standard_normal((d, d)); M = A @ A.T / d + np.eye(d)
for algo, C in (("hmc", 65536), ("mala", 65536), ("rwmh", 65536), ("nuts", 16384)):
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
    for name, kw in (("plain", {}), ("dense precond", dict(precond_mat=M))):
        st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100, n_leap_steps=16,
                                       step_size=(0.05 if algo == "hmc" else 0.3), **kw)
        t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
        for rep in range(2):
            theta = theta0.clone()
            ch = mcmc_amd.make_chains(theta, C, mem=mcmc_amd.MEM_DEVICE)
            torch.cuda.synchronize(); t0 = time.time()
            mcmc_amd.run(algo, t, st, ch); torch.cuda.synchronize()
            ms = (time.time() - t0) * 1e3
        print("%-5s %6d chains %-14s %8.1f ms" % (algo, C, name, ms))
