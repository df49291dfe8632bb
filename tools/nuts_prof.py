"""Phase clocks of the NUTS kernels (one wave of workgroup 0): `make -C mcmc_amd/csrc prof`, then on the GPU box
   MI_MCMC_LIB=mcmc_amd/libmi_mcmc_prof.so MI_NUTS_PROF=1 [MI_NUTS_HINT=9 for the tick-local kernel] python tools/nuts_prof.py"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C, d = int(os.environ.get("MI_C", "16384")), 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE, kernel_hint=int(os.environ.get("MI_NUTS_HINT", "0")))
ch = mcmc_amd.make_chains(theta, C, mem=mcmc_amd.MEM_DEVICE)
mcmc_amd.run("nuts", t, st, ch)
torch.cuda.synchronize()
