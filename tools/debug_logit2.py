import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
from test_gpu_parity_mala import _blocks
def run(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    nb, bs = _blocks(d)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=nb, block_size=bs)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=eps, W=4, hoist=1, blocks=nb, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=3)
    return g_draws, o_draws
a1, b1 = run(64, 100, 20, 0.05, 3, 8); print("fresh 64:", np.array_equal(a1, b1))
a0, b0 = run(5, 40, 16, 0.10, 5, 20); print("5:", np.array_equal(a0, b0))
a2, b2 = run(64, 100, 20, 0.05, 3, 8); print("after 5 -> 64:", np.array_equal(a2, b2), "gpu same as fresh", np.array_equal(a1, a2), "oracle same as fresh", np.array_equal(b1, b2))
