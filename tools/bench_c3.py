"""BASELINE config 3: MALA on d=512 Bayesian logistic regression (N=1024 rows), many chains, device-resident."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
burn = keep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
d, N = 512, 1024
X, y = synth.logistic_problem(d, N)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
theta0 = torch.zeros((d, C), dtype=torch.float64, device=dev)
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
nacc = torch.zeros(C, dtype=torch.int64, device=dev)
t = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, X=Xd, y=yd, mem=mcmc_amd.MEM_DEVICE)
t.n_rows = N
st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, step_size=0.02)
ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
for rep in range(2):
    theta.copy_(theta0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run("mala", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    evals = C * (burn + keep + 1)
    print(json.dumps({"config": "C3", "chains": C, "draws": burn + keep, "ms": ms, "units_per_s(chain*dim*draw)": C * d * (burn + keep) / (ms * 1e-3),
                      "draws_per_s": C * (burn + keep) / (ms * 1e-3), "TFLOPs_alg(4Nd per eval)": evals * 4.0 * N * d / (ms * 1e-3) / 1e12,
                      "accept": float(nacc.double().mean()) / keep}))
