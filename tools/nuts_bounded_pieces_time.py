"""nuts with settings.vals_bound on the built-in dense Gaussian (nuts_tile_kernel<built-in Gaussian ., true>, nuts_bounded_launch.hip) with more chains than the
16 384 chain slots of its persistent grid: the run whose pieces migrate between slots.  MI_MCMC_LIB selects the library (A/B against an older build).
python tools/nuts_bounded_pieces_time.py [chains] [draws per half]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
half = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = 128
dev = torch.device("cuda", 0)
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(np.clip(synth.initial_states(C, d, seed=3), -1.0, 1.5).T)).to(dev)
lb = np.where(np.arange(d) % 3 == 0, -1.5, -np.inf); ub = np.where(np.arange(d) % 4 == 0, 2.0, np.inf)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=half, n_keep_draws=half, n_adapt_draws=half, vals_bound=1, lower_bounds=lb, upper_bounds=ub)
t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
draws = torch.empty((half, d, C), dtype=torch.float64, device=dev)
for rep in range(2):
    theta = theta0.clone()
    ch = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE)
    torch.cuda.synchronize(); t0 = time.time()
    mcmc_amd.run("nuts", t, st, ch); torch.cuda.synchronize()
    print("%s: bounded nuts, %d chains x %d draws: %.1f ms, checksum %.17g, %s" % (os.path.basename(os.environ.get("MI_MCMC_LIB", "libmi_mcmc.so")), C, 2 * half, (time.time() - t0) * 1e3, float(draws.sum().item()), mcmc_amd.last_kernel()), flush=True)
