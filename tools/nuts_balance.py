"""How unevenly NUTS work is spread over the 16 chains of a wave (BASELINE configs[3]): executed leapfrogs per chain from the depth
trace, adaptation and sampling halves, natural grouping vs chains grouped by adapted step size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd
from mcmc_amd import synth
C, d = 16384, 128
prec = synth.dense_gaussian_precision(d)
init = synth.initial_states(C, d, seed=3)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100)
draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, keep_draws=False) if 'keep_draws' in mcmc_amd.nuts.__code__.co_varnames else mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
depth = np.asarray(g["depth"])           # [draw][chain]
work = (2.0 ** depth) - 1.0              # upper bound of the leapfrogs of a draw that reached this depth
def waves(w, order=None):
    w = w if order is None else w[order]
    grp = w.reshape(-1, 16)
    return grp.max(axis=1).sum() / grp.mean(axis=1).sum()
for name, sl in (("adaptation", slice(0, 100)), ("sampling", slice(100, 200)), ("all", slice(0, 200))):
    w = work[sl].sum(axis=0)
    print(name, "mean/chain %.0f  std %.0f  max %.0f  wave max/mean %.3f" % (w.mean(), w.std(), w.max(), waves(w)),
          " sorted by eps: %.3f" % waves(w, np.argsort(g["eps"])), " sorted by work: %.3f" % waves(w, np.argsort(w)))
print("eps quantiles", np.quantile(g["eps"], [0, .1, .5, .9, 1]))
print("depth histogram (sampling)", np.bincount(depth[100:].ravel().astype(int)))
print("n_leap total", g["n_leap"].sum(), "bound", work.sum())
