import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
from test_gpu_parity_hmc import _oracle_many
for d in (8, 128):
    prec = synth.dense_gaussian_precision(d, seed=5)
    for C in (1, 2, 3, 5, 16, 17):
        init = synth.initial_states(C, d, seed=13)
        st = mcmc_amd.default_settings(rng_seed_value=31, n_burnin_draws=1, n_keep_draws=3, n_leap_steps=2, step_size=0.1)
        g, gi = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
        o, oi = _oracle_many(orc.TARGET_DENSE, d, init, st, prec=prec)
        print("d", d, "C", C, "equal", np.array_equal(g, o), "first-draw equal", np.array_equal(g[0], o[0]), "acc", gi["n_accept"], oi["n_accept"])
