"""bench.py's extra leg `nuts_on_configs2_target` alone (mcmc::nuts on configs[2]'s target: d = 512 logistic regression, N = 1024, 32 768 chains,
4 + 4 draws, max_tree_depth 10): ms, leapfrogs as the reference counts them and as executed (GPU box): python tools/nuts_lds_extra_time.py [chains]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
d, n_rows, C, burn, keep = 512, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 4, 4
X, y = synth.logistic_problem(d, n_rows)
dev = torch.device("cuda", 0)
theta0 = torch.from_numpy(np.ascontiguousarray((synth.initial_states(C, d, seed=3) * 0.1).T)).to(dev)
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
n_leap = torch.zeros(C, dtype=torch.int64, device=dev); n_exec = torch.zeros(C, dtype=torch.int64, device=dev)
target = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, mem=mcmc_amd.MEM_DEVICE, X=torch.from_numpy(X).to(dev), y=torch.from_numpy(y).to(dev))
settings = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=burn, max_tree_depth=10, step_size=0.03)
chains = mcmc_amd.make_chains(theta, C, draws=draws, n_leapfrogs=n_leap, n_leapfrogs_executed=n_exec, step_size=torch.zeros(C, dtype=torch.float64, device=dev), mem=mcmc_amd.MEM_DEVICE)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    theta.copy_(theta0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); mcmc_amd.run("nuts", target, settings, chains, stream=stream); ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    leaps, execd = float(n_leap.double().sum()), float(n_exec.double().sum())
    print(f"{mcmc_amd.last_kernel()}: {ms:.1f} ms, {leaps:.4g} leapfrogs counted, {execd:.4g} executed ({execd / leaps:.3f}), {execd * 4 * n_rows * d / ms / 1e9:.2f} TFLOP/s executed, "
          f"{leaps * 4 * n_rows * d / ms / 1e9:.2f} reference-equivalent, checksum {float(draws[-1].sum()):.17g}", flush=True)
