"""hmc and mala with a DENSE precond_mat beyond d = 128: the LDS-streamed kernel (logistic_lds.hpp, DENSEM) against literal_kernel<0>, which served this
case before round 5 (GPU box): python tools/dense_m_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
NOLIT = os.environ.get("MI_DM_NOLIT") == "1"      # (under rocprofv3: the streamed kernels only)
from mcmc_amd import synth

def run(kind, d, Cn, L, nd, hint, N=1024, algo="hmc"):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))
    theta = torch.from_numpy(np.ascontiguousarray((synth.initial_states(Cn, d, seed=3) * 0.3).T)).cuda()
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=L, step_size=0.03, precond_mat=M)
    draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
    ch = mcmc_amd.make_chains(theta, Cn, draws=draws, mem=mcmc_amd.MEM_DEVICE)
    if kind == "dense":
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(synth.dense_gaussian_precision(d)).cuda(), mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        flop_leap = 2 * (2 * d * d)                      # P x and Minv p
    else:
        X, y = synth.logistic_problem(d, N, seed=5)
        tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=torch.from_numpy(X).cuda(), y=torch.from_numpy(y).cuda(), mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        flop_leap = 4 * N * d + 2 * d * d
    mcmc_amd.run(algo, tgt, st, ch)                      # warm-up (first launch of the instantiation)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc_amd.run(algo, tgt, st, ch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = float(Cn) * nd * (L * flop_leap + 3 * 2 * d * d)   # + L z and the two kinetic products per draw
    if algo == "mala": fl = float(Cn) * nd * ((flop_leap - 2 * d * d) + 4 * 2 * d * d)      # one evaluation + M grad, L z, two INV(Sigma) products
    # ... and a call that finds nothing memoised: a NEW precond_mat (INV / CHOL_LOWER on the device, round 6: linalg_device.hip)
    st2 = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, n_leap_steps=L, step_size=0.03, precond_mat=M * 1.0009765625)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc_amd.run(algo, tgt, st2, ch)
    torch.cuda.synchronize(); dt_new = time.perf_counter() - t0
    print(f"{algo} {kind} d={d} C={Cn} L={L} draws={nd}: {dt * 1e3:.1f} ms (new precond_mat: {dt_new * 1e3:.1f} ms), kernel {mcmc_amd.last_kernel()}, {fl / dt / 1e12:.2f} TFLOP/s algorithmic"
          f" ({fl / dt_new / 1e12:.2f} with the factorisations)", flush=True)
    return dt

for kind, d, Cn, L, nd in [("dense", 256, 8192, 16, 20), ("dense", 512, 8192, 16, 10), ("logit", 512, 8192, 8, 10), ("dense", 256, 65536, 16, 20)]:
    a = run(kind, d, Cn, L, nd, mcmc_amd.KERNEL_AUTO)
    if Cn <= 8192 and not NOLIT:
        b = run(kind, d, Cn, L, nd, mcmc_amd.KERNEL_LITERAL)
        print(f"   -> {b / a:.1f}x the literal kernel")

for kind, d, Cn, nd in [("dense", 256, 8192, 40), ("logit", 512, 8192, 20)]:
    a = run(kind, d, Cn, 0, nd, mcmc_amd.KERNEL_AUTO, algo="mala")
    if NOLIT: continue
    b = run(kind, d, Cn, 0, nd, mcmc_amd.KERNEL_LITERAL, algo="mala")
    print(f"   -> {b / a:.1f}x the literal kernel")
