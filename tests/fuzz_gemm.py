"""Randomised sweep of the matrix-product samplers (mcmc_amd/csrc/gemm_samplers.hip: hmc / mala / rwmh beyond d = 512, dense Gaussians and the logistic
target) against the literal kernels of the same library (MI_KERNEL_LITERAL: one workgroup per chain, the reference's operations as written, themselves
pinned against the oracle by tests/test_gpu_literal_paths.py and the CPU suite) -- both run on the GPU: ragged d and N (not multiples of 16 / 128),
ragged chain tiles, a diagonal precond_mat, 0 .. many draws, one .. several leapfrog steps, step sizes from tiny to absurd, chains that start in the non-finite regime,
chain0 / draw0 offsets, runs cut in two.  Bit-exact or report.
Usage (GPU box): python tests/fuzz_gemm.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mcmc_amd
from mcmc_amd import synth


def sweep(n_cases=30, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(n_cases):
        algo = str(rng.choice(["hmc", "hmc", "mala", "rwmh"]))
        kind = str(rng.choice(["logistic", "dense"]))
        d = int(rng.choice([513, 520, 528, 600, 640, 641, 767, 768, 1000, 1024, 1025, 1153]))
        if kind == "logistic":
            n_rows = int(rng.choice([1, 7, 16, 17, 100, 128, 129, 300]))
            X, y = synth.logistic_problem(d, n_rows, seed=int(rng.integers(1, 99)))
            tk, tkw, scale = mcmc_amd.TARGET_LOGISTIC, dict(X=X, y=y), 0.1
        else:
            n_rows = 0
            tk, tkw, scale = mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=synth.dense_gaussian_precision(d, seed=int(rng.integers(1, 99)))), 0.5
        C = int(rng.choice([1, 5, 16, 64, 127, 128, 129, 200, 300]))
        burn, keep = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        if burn + keep == 0: keep = 1
        L = int(rng.choice([1, 2, 3, 5]))
        eps = float(rng.choice([0.002, 0.01, 0.03, 0.1, 1.0, 1e5]))
        init = synth.initial_states(C, d, seed=int(rng.integers(1, 1000))) * scale
        wild = rng.random() < 0.3
        if wild:
            for c in rng.choice(C, size=min(C, 3), replace=False):
                init[c] *= float(rng.choice([1e150, 1e300]))
                if rng.random() < 0.3: init[c, int(rng.integers(0, d))] = float(rng.choice([np.inf, -np.inf, np.nan]))
        sd = int(rng.integers(1, 10**6))
        chain0, draw0 = int(rng.integers(0, 5000)), int(rng.choice([0, 0, 3]))
        kw = {}
        if algo != "rwmh" and rng.random() < 0.35: kw["precond_mat"] = np.diag(rng.uniform(0.3, 3.0, d))      # a DIAGONAL precond_mat (hmc, mala)
        S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=sd, n_burnin_draws=b, n_keep_draws=k, n_leap_steps=L, step_size=eps, **kw)
        a_draws, a = mcmc_amd.sample(algo, tk, init, S(burn, keep), chain0=chain0, draw0=draw0, **tkw)
        kernel = mcmc_amd.last_kernel()
        b_draws, b = mcmc_amd.sample(algo, tk, init, S(burn, keep), chain0=chain0, draw0=draw0, kernel_hint=mcmc_amd.KERNEL_LITERAL, **tkw)
        bits = lambda v: np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
        same = lambda u, v: np.array_equal(bits(u), bits(v)) or np.array_equal(u, v, equal_nan=True)     # (NaN payloads may differ)
        ok = (kernel.startswith("gemm_step_kernel<") and mcmc_amd.last_kernel().startswith("literal_kernel<")
              and same(a_draws, b_draws) and np.array_equal(a["n_accept"], b["n_accept"]) and same(a["theta"], b["theta"]) and np.array_equal(a["n_leap"], b["n_leap"]))
        cut = None
        if ok and not wild and burn == 0 and keep >= 2 and draw0 == 0:       # the same run cut in two (all draws kept: rows compare one to one)
            cut = int(rng.integers(1, keep))
            p_draws, p = mcmc_amd.sample(algo, tk, init, S(0, cut), chain0=chain0, **tkw)
            q_draws, q = mcmc_amd.sample(algo, tk, p["theta"].T.copy(), S(0, keep - cut), chain0=chain0, draw0=cut, **tkw)
            ok = same(np.concatenate([p_draws, q_draws]), a_draws) and np.array_equal(p["n_accept"] + q["n_accept"], a["n_accept"])
        if verbose or not ok:
            print(("ok  " if ok else "FAIL"), dict(algo=algo, kind=kind, d=d, n_rows=n_rows, C=C, burn=burn, keep=keep, L=L, eps=eps, chain0=chain0, draw0=draw0, wild=wild, diag=bool(kw), cut=cut,
                                                   seed=sd, kernel=kernel, acc=int(a["n_accept"].sum())), flush=True)
        fails += 0 if ok else 1
    return fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = sweep(n, s)
    print("mismatching cases:", f)
    sys.exit(1 if f else 0)
