"""mcmc::rwmh many-chain throughput: d=128 dense Gaussian, 65 536 chains, device-resident, HIP-event timed."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C, d, burn, keep = 65536, 128, 100, 100
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
nacc = torch.zeros(C, dtype=torch.int64, device=dev)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, step_size=0.05)
ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
for rep in range(2):
    theta.copy_(theta0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run("rwmh", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
print(json.dumps({"algo": "rwmh", "chains": C, "d": d, "draws": burn + keep, "ms": ms, "proposals_per_s": C * (burn + keep) / (ms * 1e-3),
                  "TFLOPs(2d^2 per draw)": C * (burn + keep) * 2.0 * d * d / (ms * 1e-3) / 1e12, "accept": float(nacc.double().mean()) / keep}))
