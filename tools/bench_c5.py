"""BASELINE config 5 shard: HMC d=1024 ill-conditioned diagonal Gaussian, 131072 chains (one GPU's share of 2^20)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
dev = torch.device("cuda", 0)
d, C, L, burn, keep = 1024, 131072, 32, 20, 8
prec = torch.from_numpy(synth.ill_conditioned_diag(d, 1e4)).to(dev)
theta0 = torch.randn((d, C), dtype=torch.float64, device=dev) / torch.sqrt(prec)[:, None]
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
nacc = torch.zeros(C, dtype=torch.int64, device=dev)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DIAG, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.005)
ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
for rep in range(3):
    theta.copy_(theta0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run("hmc", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    units = C * d * L * (burn + keep)
    print(json.dumps({"config": "C5 shard", "ms": ms, "units_per_s": units / (ms * 1e-3), "accept": float(nacc.double().mean()) / keep,
                      "valu_fp64_ops_per_s": units * 7 / (ms * 1e-3)}))
