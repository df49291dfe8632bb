"""Cost of a diagonal precond_mat on the LDS-streamed kernels (logit_lds_kernel<., ., ., true>) next to the identity: python tools/logit_diag_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
d, N, Cn = 512, 1024, 65536
X, y = synth.logistic_problem(d, N, seed=4)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
M = np.diag(np.linspace(0.5, 2.0, d))
for algo, tgt, nd, L in [("hmc", "logit", 10, 8), ("mala", "logit", 40, 0), ("hmc", "dense", 10, 16), ("mala", "dense", 100, 0)]:
    for name, kw in [("identity", {}), ("diagonal precond_mat", dict(precond_mat=M))]:
        theta = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T * 0.1)).cuda()
        st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=max(L, 1), step_size=0.02, **kw)
        ch = mcmc_amd.make_chains(theta, Cn, mem=mcmc_amd.MEM_DEVICE)
        t = (mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=Xd, y=yd, mem=mcmc_amd.MEM_DEVICE) if tgt == "logit"
             else mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE))
        best = 1e9
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mcmc_amd.run(algo, t, st, ch)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{algo} {tgt} d={d} C={Cn} {nd} draws: {name:22s} {best * 1e3:8.1f} ms  {mcmc_amd.last_kernel()}", flush=True)
