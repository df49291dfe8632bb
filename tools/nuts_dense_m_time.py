"""nuts with a DENSE precond_mat beyond d = 128 / on the logistic target: the LDS-streamed kernel (nuts_lds.hpp, DENSEM, round 6) against literal_kernel<2>,
which served this case until round 5 (GPU box): python tools/nuts_dense_m_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth


def run(kind, d, Cn, depth, nd, hint, N=1024):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))
    theta = torch.from_numpy(np.ascontiguousarray((synth.initial_states(Cn, d, seed=3) * 0.3).T)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_adapt_draws=nd // 2, max_tree_depth=depth, step_size=0.03, precond_mat=M)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    nl = torch.zeros(Cn, dtype=torch.int64, device="cuda")
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_leapfrogs=nl, mem=mcmc_amd.MEM_DEVICE)
    if kind == "dense":
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(synth.dense_gaussian_precision(d)).cuda(), mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        flop_leap = 2 * d * d + 2 * (2 * d * d)          # P x, and Minv p twice (drift, kinetic energy)
    else:
        X, y = synth.logistic_problem(d, N, seed=5)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=torch.from_numpy(X).cuda(), y=torch.from_numpy(y).cuda(), mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        flop_leap = 4 * N * d + 2 * (2 * d * d)
    mcmc_amd.run("nuts", tgt, st, ch)                    # warm-up (first launch of the instantiation)
    theta.copy_(torch.from_numpy(np.ascontiguousarray((synth.initial_states(Cn, d, seed=3) * 0.3).T)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc_amd.run("nuts", tgt, st, ch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    leaps = float(nl.sum().item())
    print(f"nuts {kind} d={d} C={Cn} depth<={depth} draws={nd}: {dt * 1e3:.1f} ms, {leaps / Cn / nd:.1f} leapfrogs per draw, kernel {mcmc_amd.last_kernel()}, "
          f"{leaps * flop_leap / dt / 1e12:.2f} TFLOP/s algorithmic", flush=True)
    return dt


for kind, d, Cn, depth, nd in [("dense", 256, 8192, 5, 10), ("dense", 512, 8192, 4, 6), ("logit", 256, 8192, 5, 10), ("logit", 512, 8192, 4, 6)]:
    a = run(kind, d, Cn, depth, nd, mcmc_amd.KERNEL_AUTO)
    b = run(kind, d, Cn, depth, nd, mcmc_amd.KERNEL_LITERAL)
    print(f"   -> {b / a:.1f}x the literal kernel", flush=True)
