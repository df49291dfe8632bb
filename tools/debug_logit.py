import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
from test_gpu_parity_mala import _blocks
for (d, N, C, keep, burn) in [(64, 100, 20, 8, 3), (64, 100, 20, 8, 0), (64, 100, 16, 8, 3), (64, 100, 16, 2, 1)]:
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=0.05)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    nb, bs = _blocks(d)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=nb, block_size=bs)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=0.05, W=4, hoist=1, blocks=nb, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=3)
    diff = np.abs(g_draws - o_draws)
    print(d, N, C, keep, burn, "acc eq", np.array_equal(g["n_accept"], o["n_accept"]), "max diff per draw", diff.max(axis=(1, 2)), "chains with diff", np.nonzero(diff.max(axis=(0,1)))[0][:24])
