"""Rate of the literal kernels (mcmc_amd/csrc/literal.hpp) as the device path beyond the tiled kernels (GPU box): python tools/literal_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
for algo, d, Cn, L, nd in [("hmc", 256, 8192, 16, 20), ("hmc", 512, 8192, 16, 10), ("nuts", 256, 4096, 0, 20), ("mala", 256, 8192, 0, 40), ("hmc", 128, 8192, 16, 20)]:
    P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
    theta = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=max(L, 1), step_size=0.03, n_adapt_draws=nd // 2,
                                   max_tree_depth=11 if (algo == "nuts") else 10)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    nleap = torch.zeros(Cn, dtype=torch.int64, device="cuda")
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_leapfrogs=nleap, step_size=torch.zeros(Cn, dtype=torch.float64, device="cuda"), mem=mcmc_amd.MEM_DEVICE)
    tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc_amd.run(algo, tgt, st, ch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    units = float(nleap.double().sum()) * d if algo != "mala" else float(Cn) * d * nd
    print(f"{algo} d={d} C={Cn}: {dt * 1e3:.1f} ms, kernel {mcmc_amd.last_kernel()}, {units / dt:.3e} units/s, {units / dt * (2 * d + 8) / 1e12:.2f} TFLOP/s algorithmic")
