import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd
from mcmc_amd import synth
def run(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    return mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)[0]
h = lambda a: hashlib.md5(a.tobytes()).hexdigest()[:8]
print("A: same call repeated, nothing in between")
import collections; print(collections.Counter([h(run(64, 100, 20, 0.05, 3, 8)) for _ in range(60)]))
