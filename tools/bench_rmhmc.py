"""mcmc::rmhmc many-chain throughput: the reference's example model (d=2 normal model, 1000 observations, Fisher
metric, n_leap=1, n_fp=5: examples/eigen/rmhmc_normal.cpp), 65 536 chains, device-resident, HIP-event timed."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
C, d, n, burn, keep, n_fp = int(os.environ.get("CHAINS", 65536)), 2, 1000, 100, 100, 5
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
x = torch.from_numpy(2.0 + 2.0 * rng.standard_normal(n)).to(dev)
init = np.stack([2.0 + rng.uniform(-0.5, 0.5, C), 2.0 + rng.uniform(-0.3, 0.8, C)])          # [d][C]
theta0 = torch.from_numpy(np.ascontiguousarray(init)).to(dev)
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
nacc = torch.zeros(C, dtype=torch.int64, device=dev)
t = mcmc_amd.make_target(mcmc_amd.TARGET_NORMAL_MODEL, d, y=x, mem=mcmc_amd.MEM_DEVICE)
t.n_rows = n
st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, step_size=0.02, n_leap_steps=1, n_fp_steps=n_fp)
ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
for rep in range(2):
    theta.copy_(theta0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run("rmhmc", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
evals = (n_fp + 1) + 1                                    # target evaluations (each a pass over the n observations) per draw
flop = C * (burn + keep) * evals * n * 4.0                # sub, add, fma per observation
print(json.dumps({"algo": "rmhmc", "chains": C, "n_obs": n, "draws": burn + keep, "ms": ms, "draws_per_s": C * (burn + keep) / (ms * 1e-3),
                  "target_evals_per_s": C * (burn + keep) * evals / (ms * 1e-3), "TFLOPs_data_sums": flop / (ms * 1e-3) / 1e12,
                  "accept": float(nacc.double().mean()) / keep}))
