import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, orc
from mcmc_amd import synth
d = 128
P = synth.dense_gaussian_precision(d)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
for nt in (1, 8, 32, 64, 128, 256):
    n = 2 * nt
    init = synth.initial_states(n, d)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P)
    s = orc.make_settings(seed=1, n_burnin=20, n_keep=20, n_leap=16, step=0.05)
    t0 = time.perf_counter()
    _, info = orc.run_many(orc.ALGO_HMC, t, init, s, n_threads=nt, want_draws=False)
    dt = time.perf_counter() - t0
    print(nt, "threads:", n, "chains", f"{dt:.2f}s", f"{info['n_leap'].sum() * d / dt:.3e} units/s")
