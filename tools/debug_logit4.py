import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd
from mcmc_amd import synth
def run(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    return mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)[0]
for (d, N, C) in [(64, 100, 20), (64, 96, 20), (64, 112, 20), (64, 100, 16), (64, 100, 32), (64, 16, 20), (16, 100, 20), (64, 100, 4096)]:
    outs = []
    for rep in range(6):
        run(5 + rep, 40 + rep, 16, 0.1, 1, 2)           # perturb allocator / timing between repetitions
        outs.append(run(d, N, C, 0.05, 3, 8))
    same = [bool(np.array_equal(outs[0], o)) for o in outs]
    print(d, N, C, "NB", (N + 15) // 16, "runs equal to first:", same)
