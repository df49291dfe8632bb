"""Rate of the LDS-streamed dense Gaussian kernel (logistic_lds.hpp, LOGIT_TARGET_DENSE; 128 < d <= 512) on the GPU box:
python tools/dense_lds_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
import os
CASES = [("hmc", 512, 65536, 16, 10), ("hmc", 256, 65536, 16, 20)] if os.environ.get("DENSE_LDS_SHORT") else [("hmc", 512, 65536, 16, 10), ("hmc", 256, 65536, 16, 20), ("hmc", 192, 65536, 16, 20), ("mala", 512, 65536, 0, 100), ("rwmh", 512, 65536, 0, 100),
                           ("hmc", 512, 8192, 16, 10)]
for algo, d, Cn, L, nd in CASES:
    P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
    theta = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=max(L, 1), step_size=0.03)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, mem=mcmc_amd.MEM_DEVICE)
    tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run(algo, tgt, st, ch)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    evals = nd * (L if algo == "hmc" else 1) + 1
    flop = 2.0 * d * d * Cn * evals
    print(f"{algo} d={d} C={Cn}: {best * 1e3:.1f} ms, kernel {mcmc_amd.last_kernel()}, {flop / best / 1e12:.2f} TFLOP/s (2 d^2 per gradient), {flop / best / 78.6e12:.3f} of the fp64 matrix peak", flush=True)
