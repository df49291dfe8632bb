"""nuts on the LDS-streamed evaluation (mcmc_amd/csrc/nuts_lds.hpp) against literal_kernel<2> on the same problem, same bits (GPU box):
python tools/nuts_lds_time.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
SHAPES = [("dense", 256, 0, 4096, 20), ("dense", 512, 0, 4096, 10), ("logistic", 512, 1024, 4096, 10), ("dense", 256, 0, 16384, 20), ("logistic", 512, 1024, 16384, 10),
          ("dense", 256, 0, 65536, 20), ("dense", 512, 0, 32768, 10), ("logistic", 512, 1024, 65536, 10)]
if quick: SHAPES = SHAPES[:3]
for kind, d, n_rows, Cn, nd in SHAPES:
    res = {}
    for name, hint in (("lds", mcmc_amd.KERNEL_AUTO), ("literal", mcmc_amd.KERNEL_LITERAL)):
        if name == "literal" and Cn > 4096: continue
        theta = torch.from_numpy(np.ascontiguousarray((synth.initial_states(Cn, d, seed=3) * (0.1 if kind == "logistic" else 1.0)).T)).cuda()
        st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=nd // 2, n_keep_draws=nd - nd // 2, step_size=0.03, n_adapt_draws=nd // 2, max_tree_depth=10)
        draws = torch.empty((nd - nd // 2, d, Cn), dtype=torch.float64, device="cuda")
        nleap = torch.zeros(Cn, dtype=torch.int64, device="cuda")
        ch = mcmc_amd.make_chains(theta, Cn, draws=draws, n_leapfrogs=nleap, step_size=torch.zeros(Cn, dtype=torch.float64, device="cuda"), mem=mcmc_amd.MEM_DEVICE)
        if kind == "dense":
            P = torch.from_numpy(synth.dense_gaussian_precision(d)).cuda()
            tgt = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=P, mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        else:
            X, y = synth.logistic_problem(d, n_rows, seed=1)
            tgt = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=torch.from_numpy(X).cuda(), y=torch.from_numpy(y).cuda(), mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mcmc_amd.run("nuts", tgt, st, ch)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        leaps = float(nleap.double().sum())
        flops = leaps * (2.0 * d * d if kind == "dense" else 4.0 * d * n_rows)
        res[name] = (dt, draws.clone(), leaps)
        print(f"nuts {kind} d={d} C={Cn} {name}: {dt * 1e3:.1f} ms, kernel {mcmc_amd.last_kernel()}, {leaps:.3e} leapfrogs, {leaps * d / dt:.3e} units/s, {flops / dt / 1e12:.2f} TFLOP/s algorithmic", flush=True)
    if "literal" in res:
        print(f"   speed-up {res['literal'][0] / res['lds'][0]:.1f}x, same draws: {bool(torch.equal(res['lds'][1], res['literal'][1]))}", flush=True)
