"""ESS/sec on the ill-conditioned diagonal Gaussian of BASELINE config 5 (d=1024, kappa=1e4), with and without a diagonal mass
matrix ("needed for ill-conditioned targets to mix"): identity precond_mat needs eps ~ 1/sqrt(lambda_max) and leaves
the low-precision dimensions almost frozen; M = diag(precision) runs every dimension at unit frequency."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
dev = torch.device("cuda", 0)
d, C, L, burn, keep = 1024, 16384, 32, 50, 100
prec_h = synth.ill_conditioned_diag(d, 1e4)
prec = torch.from_numpy(prec_h).to(dev)
theta0 = torch.randn((d, C), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) / torch.sqrt(prec)[:, None]
for name, eps, M in (("identity", 0.005, None), ("M=diag(prec)", 0.12, np.diag(prec_h))):
    theta = theta0.clone()
    draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
    nacc = torch.zeros(C, dtype=torch.int64, device=dev)
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DIAG, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
    kw = {} if M is None else dict(precond_mat=M)
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps, **kw)
    ch = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mcmc_amd.run("hmc", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = mcmc_amd.draw_stats(draws, keep, d, C, mem=mcmc_amd.MEM_DEVICE)
    print(json.dumps({"precond": name, "eps": eps, "ms": ms, "accept": float(nacc.double().mean()) / keep,
                      "ess_min_per_chain": float(s["ess"].min()), "ess_median_per_chain": float(np.median(s["ess"])),
                      "ess_per_sec(min over dims x chains)": float(s["ess"].min()) * C / (ms * 1e-3), "rhat_max": float(s["rhat"].max())}))
