import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd
from mcmc_amd import synth
d, C = 128, 16384
prec = synth.dense_gaussian_precision(d)
init = synth.initial_states(C, d, seed=3)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100)
t0 = time.time(); draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, want_draws=False); dt = time.time() - t0
depth = g["depth"].astype(np.int64)            # [n_tot, C]
leaps_chain = (2 ** depth - 1)                 # upper bound of leapfrogs per chain-draw
wave_depth = depth.reshape(depth.shape[0], C // 16, 16).max(axis=2)
ticks_wave = (2 ** wave_depth - 1).sum()
print("host wall s", dt, "mean depth", depth.mean(), "mean leaps/chain-draw (bound)", leaps_chain.mean(), "actual", g["n_leap"].mean() / 200)
print("wave ticks total", ticks_wave, "per wave-draw", ticks_wave / wave_depth.size, "utilisation = actual leaps / (16*ticks):", g["n_leap"].sum() / (16 * ticks_wave))
print("depth histogram", np.bincount(depth.ravel()))
