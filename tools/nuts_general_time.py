import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C, d = 16384, 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
M = np.diag(np.linspace(0.5, 2.0, d))
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100, precond_mat=M)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
for rep in range(2):
    theta = theta0.clone()
    ch = mcmc_amd.make_chains(theta, C, mem=mcmc_amd.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.time()
    mcmc_amd.run("nuts", t, st, ch); torch.cuda.synchronize()
    print("general nuts (diag precond) 16384 chains: %.1f ms" % ((time.time() - t0) * 1e3))
