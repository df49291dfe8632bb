"""Throughput of the other BASELINE configs (not the bench.py line): device-resident, HIP-event timed."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
import mcmc_amd
from mcmc_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="nuts")
ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--burn", type=int, default=100)
ap.add_argument("--keep", type=int, default=100)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--leap", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda", 0)
d, C = args.d, args.chains
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
theta = torch.empty_like(theta0)
draws = torch.empty((args.keep, d, C), dtype=torch.float64, device=dev)
n_accept = torch.zeros(C, dtype=torch.int64, device=dev)
n_leap = torch.zeros(C, dtype=torch.int64, device=dev)
eps = torch.zeros(C, dtype=torch.float64, device=dev)
target = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
if args.algo == "nuts":
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=args.burn, n_keep_draws=args.keep, n_adapt_draws=args.burn)
else:
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=args.burn, n_keep_draws=args.keep, n_leap_steps=args.leap, step_size=0.05)
ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=n_accept, n_leapfrogs=n_leap, step_size=eps, mem=mcmc_amd.MEM_DEVICE)
stream = torch.cuda.current_stream().cuda_stream
for rep in range(args.reps):
    theta.copy_(theta0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run(args.algo, target, st, ch, stream=stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    leaps = float(n_leap.double().sum().item())
    print(json.dumps({"algo": args.algo, "chains": C, "d": d, "ms": ms, "leapfrogs_executed": leaps,
                      "units_per_s": leaps * d / (ms * 1e-3), "mean_leaps_per_draw": leaps / C / (args.burn + args.keep),
                      "max_over_mean_leaps": float(n_leap.max().item()) / (leaps / C),
                      "accept": float(n_accept.double().mean().item()) / args.keep, "n_leap0": int(n_leap[0].item()), "eps_mean": float(eps.mean().item())}))
