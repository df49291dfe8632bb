"""Rate of the matrix-product samplers (gemm_samplers.hip: dense Gaussian targets beyond d = 512) on the GPU box, next to the literal kernel they replace:
python tools/gemm_time.py [short]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
short = len(sys.argv) > 1 and sys.argv[1] == "short"
CASES = [("hmc", 1024, 65536, 16, 4)] if short else [("hmc", 1024, 65536, 16, 6), ("hmc", 2048, 32768, 16, 4), ("hmc", 640, 65536, 16, 6), ("hmc", 1024, 8192, 16, 6),
                                                      ("hmc", 1024, 1024, 16, 6), ("mala", 1024, 65536, 0, 40), ("rwmh", 1024, 65536, 0, 40), ("hmc", 4096, 8192, 8, 2)]
for algo, d, Cn, L, nd in CASES:
    P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=max(L, 1), step_size=0.02)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    nacc = torch.zeros(Cn, dtype=torch.int64, device="cuda")
    best = 1e9
    for rep in range(3):
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run(algo, tgt, st, ch)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    evals = nd * (L if algo == "hmc" else 1) + 1
    flop = 2.0 * d * d * Cn * evals
    print(f"{algo} d={d} C={Cn} L={L} draws={nd}: {best * 1e3:.1f} ms, kernel {mcmc_amd.last_kernel()}, {flop / best / 1e12:.2f} TFLOP/s (2 d^2 per gradient), "
          f"{flop / best / 78.6e12:.3f} of the fp64 matrix peak, accept rate {nacc.double().mean().item() / max(nd - nd // 2, 1):.2f}", flush=True)
    if Cn <= 1024 and not short:      # the literal kernel it replaces, same call
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE, kernel_hint=mcmc_amd.KERNEL_LITERAL)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run(algo, tgt, st, ch)
        torch.cuda.synchronize(); tl = time.perf_counter() - t0
        print(f"    {mcmc_amd.last_kernel()}: {tl * 1e3:.1f} ms ({tl / best:.1f}x)", flush=True)

# the logistic-regression target beyond d = 512: two products per gradient (eta = X Theta, X^T (y - sigmoid(eta))), 4 N d flop
LCASES = [("hmc", 1024, 1024, 65536, 8, 4)] if short else [("hmc", 1024, 1024, 65536, 8, 4), ("hmc", 2048, 512, 32768, 8, 4), ("mala", 1024, 1024, 65536, 0, 20), ("hmc", 1024, 1024, 1024, 8, 4)]
for algo, d, N, Cn, L, nd in LCASES:
    Xh, yh = synth.logistic_problem(d, N, seed=5)
    X = torch.from_numpy(Xh).cuda(); y = torch.from_numpy(yh).cuda()
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(Cn, d, seed=3).T * 0.1)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=max(L, 1), step_size=0.01)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    nacc = torch.zeros(Cn, dtype=torch.int64, device="cuda")
    best = 1e9
    for rep in range(3):
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=X, y=y, mem=mcmc_amd.MEM_DEVICE)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run(algo, tgt, st, ch)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    evals = nd * (L if algo == "hmc" else 1) + 1
    flop = 4.0 * N * d * Cn * evals
    print(f"logistic {algo} d={d} N={N} C={Cn} L={L} draws={nd}: {best * 1e3:.1f} ms, kernel {mcmc_amd.last_kernel()}, {flop / best / 1e12:.2f} TFLOP/s (4 N d per gradient), "
          f"{flop / best / 78.6e12:.3f} of the fp64 matrix peak, accept rate {nacc.double().mean().item() / max(nd - nd // 2, 1):.2f}", flush=True)
    if Cn <= 1024 and not short:
        theta = theta0.clone()
        ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=X, y=y, mem=mcmc_amd.MEM_DEVICE, kernel_hint=mcmc_amd.KERNEL_LITERAL)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run(algo, tgt, st, ch)
        torch.cuda.synchronize(); tl = time.perf_counter() - t0
        print(f"    {mcmc_amd.last_kernel()}: {tl * 1e3:.1f} ms ({tl / best:.1f}x)", flush=True)
