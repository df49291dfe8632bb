import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
from test_gpu_parity_nuts import _oracle
d, C = 8, 16
prec = synth.dense_gaussian_precision(d, seed=6)
init = synth.initial_states(C, d, seed=21)
for (burn, keep, adapt, md, eps0) in [(0, 4, 0, 3, 0.1), (0, 4, 0, 4, 0.1), (0, 4, 0, 5, 0.1), (0, 4, 0, 10, 0.3)]:
    st = mcmc_amd.default_settings(rng_seed_value=77, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=adapt, max_tree_depth=md, step_size=eps0)
    g_draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec, chain0=500)
    o_draws, o = _oracle(orc.TARGET_DENSE, d, init, st, prec=prec, chain0=500)
    print("case", burn, keep, adapt, md, eps0, {k: bool(np.array_equal(g[k], o[k])) for k in ("depth", "n_leap", "n_accept", "eps")})
    for c in range(C):
        for i in range(keep):
            if not np.array_equal(g_draws[i, :, c], o_draws[i, :, c]) or g["depth"][i, c] != o["depth"][i, c]:
                print(f"   chain {c}: first mismatch at draw {i}: depth gpu {g['depth'][:, c]} orc {o['depth'][:, c]} nleap {g['n_leap'][c]} {o['n_leap'][c]}")
                break
