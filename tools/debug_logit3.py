import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
import test_gpu_parity_mala as T
def run(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    nb, bs = T._blocks(d)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=nb, block_size=bs)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=eps, W=4, hoist=1, blocks=nb, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=3)
    return g_draws, o_draws, g, o
for c in T.CASES:
    T.test_mala_bit_exact_vs_oracle(*c)
T.test_mala_rejects_and_samples_the_target()
g, o, gi, oi = run(5, 40, 16, 0.10, 5, 20); print("d=5 ok", np.array_equal(g, o))
g, o, gi, oi = run(64, 100, 20, 0.05, 3, 8)
print("d=64: gpu==oracle", np.array_equal(g, o), "acc eq", np.array_equal(gi["n_accept"], oi["n_accept"]))
diff = np.abs(g - o)
print(" draws with diff", np.nonzero(diff.max(axis=(1,2)))[0], "chains", np.nonzero(diff.max(axis=(0,1)))[0], "dims", np.nonzero(diff.max(axis=(0,2)))[0][:70])
g2, o2, _, _ = run(64, 100, 20, 0.05, 3, 8)
print("again: gpu==oracle", np.array_equal(g2, o2), "gpu same as before", np.array_equal(g, g2), "oracle same as before", np.array_equal(o, o2))
