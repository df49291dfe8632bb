"""Stress of the piece hand-over of nuts_gauss_memo_kernel (nuts_memo_core.hpp, SPLIT: pieces of a chain's run migrate between chain slots -- and XCDs -- through memory
inside ONE launch): 65 536 chains, a few of them started non-finite, short runs repeated; every repetition must reproduce the bits of the tick-local kernel
(nuts_async.hpp: whole chains in fixed slots, an independent implementation).  python tools/nuts_pieces_stress.py [reps] [draws per half]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
half = int(sys.argv[2]) if len(sys.argv) > 2 else 16
C, d = 65536, 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
init = synth.initial_states(C, d, seed=3)
init[5] *= 1e300; init[20000, 7] = np.inf; init[40000, 100] = np.nan; init[65535] *= 1e160
theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=half, n_keep_draws=half, n_adapt_draws=half)
def run(hint):
    theta = theta0.clone()
    draws = torch.empty((half, d, C), dtype=torch.float64, device=dev)
    n_leap = torch.zeros(C, dtype=torch.int64, device=dev); eps = torch.zeros(C, dtype=torch.float64, device=dev); nacc = torch.zeros(C, dtype=torch.int64, device=dev)
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
    ch = mcmc_amd.make_chains(theta, C, draws=draws, n_leapfrogs=n_leap, step_size=eps, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
    mcmc_amd.run("nuts", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    return draws, theta, n_leap, eps, nacc, mcmc_amd.last_kernel()
ref = run(mcmc_amd.KERNEL_NUTS_TICK_LOCAL)
print("reference:", ref[5], "leapfrogs", int(ref[2].sum().item()), flush=True)
bad = 0
eq = lambda a, b: bool(torch.equal(a.view(torch.int64), b.view(torch.int64)) or torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0)))
for r in range(reps):
    got = run(mcmc_amd.KERNEL_AUTO)
    ok = eq(got[0], ref[0]) and eq(got[1], ref[1]) and bool(torch.equal(got[2], ref[2])) and eq(got[3], ref[3]) and bool(torch.equal(got[4], ref[4]))
    bad += 0 if ok else 1
    print(("ok  " if ok else "FAIL"), r, got[5], flush=True)
print("mismatching repetitions:", bad)
sys.exit(1 if bad else 0)
