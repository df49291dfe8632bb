"""configs[1]'s shape with PER-CHAIN diagonal masses (mi_chains.mass_diag) on the MFMA kernel (hmc_gauss_mfma_kernel<8, 8, false, false, true, true>)
next to the plain run.  GPU box: python tools/chain_mass_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C, d = 65536, 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
mass = torch.from_numpy(np.random.default_rng(1).uniform(0.5, 2.0, (d, C))).to(dev)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_leap_steps=16, step_size=0.05)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
draws = torch.empty((100, d, C), dtype=torch.float64, device=dev)
for name, kw in (("plain", {}), ("per-chain masses", dict(mass_diag=mass))):
    for rep in range(2):
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE, **kw)
        torch.cuda.synchronize(); t0 = time.time()
        mcmc_amd.run("hmc", t, st, ch); torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
    print(f"{name}: {ms:.1f} ms ({mcmc_amd.last_kernel()})")
