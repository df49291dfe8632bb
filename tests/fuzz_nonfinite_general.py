"""The general variants (bounds, a diagonal or dense precond_mat / cov_mat) x every Gaussian target x a dimension that does not fill its 16-wide
tiles x chains that START non-finite: the corner the round-5 fuzz caught in mala (mala_gauss_dense_m_kernel, d = 31, a chain started at -inf: the zero rows of the PADDING dimensions turn 0 * inf into
NaN, and the next dense product spread it over every real dimension through the zero padded column -- draws NaN where the reference has
+-inf).  Every sampler with real dense products on padded tiles, against the oracle.  GPU: python tests/fuzz_nonfinite_general.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import mcmc_amd
import orc
from mcmc_amd import synth


def sweep(n_cases=48, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(n_cases):
        algo = ["hmc", "mala", "nuts", "rwmh"][case % 4]
        d = int(rng.choice([5, 16, 17, 31, 32, 33, 47, 63, 64, 65, 100, 127, 128]))
        if algo == "mala" and d > 64: d = int(rng.choice([17, 31, 47, 63]))
        C = int(rng.choice([3, 16, 20]))
        rseed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
        tgt = str(rng.choice(["dense", "diag", "iso"]))
        prec, kg, ko = {"dense": (synth.dense_gaussian_precision(d, seed=rseed % 97), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE),
                        "diag": (synth.ill_conditioned_diag(d, 20.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG),
                        "iso": (None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO)}[tgt]
        gen = str(rng.choice(["dense_m", "diag_m", "bounds", "bounds+diag_m"] if algo != "rwmh" else ["dense_m", "diag_m"]))
        kw, okw = {}, {}
        if "bounds" in gen:
            kind = rng.integers(1, 5, d)
            lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
            kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
        if "_m" in gen:
            M = np.diag(rng.uniform(0.3, 3.0, d))
            if gen == "dense_m": A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + M
            kw.update(precond_mat=M); okw.update(precond=M)
        init = synth.initial_states(C, d, seed=rseed % 1013) * float(rng.choice([0.1, 1.0]))
        for _ in range(int(rng.integers(1, 3))):            # one or two non-finite starting coordinates, in one or two chains
            init[int(rng.integers(0, C)), int(rng.integers(0, d))] = float(rng.choice([np.inf, -np.inf, np.nan, 1e300, -1e308]))
        eps = float(rng.choice([0.05, 0.5, 1.0e5, 1.0e160]))
        burn, keep, L = int(rng.integers(0, 3)), int(rng.integers(1, 5)), int(rng.integers(1, 5))
        depth = int(rng.integers(1, 6))
        st = mcmc_amd.default_settings(rng_seed_value=rseed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps,
                                       n_adapt_draws=burn, max_tree_depth=depth, **kw)
        s = orc.make_settings(seed=rseed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, n_adapt=burn, max_depth=depth, W=4, hoist=1, **okw)
        t = orc.TargetSpec(ko, d, prec=prec, W=4)
        desc = f"{algo} {tgt} {gen} d={d} C={C} eps={eps} L={L} burn={burn} keep={keep} depth={depth}"
        try:
            g_draws, g = mcmc_amd.sample(algo, kg, init, st, prec=prec, chain0=chain0)
        except mcmc_amd.MiMcmcError as e:
            if verbose: print("REFUSED", desc, str(e)[:80])
            continue
        o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH}[algo], t, init, s, chain0=chain0)
        ok = np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["n_accept"], o["n_accept"])
        if algo == "nuts": ok = ok and np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
        if not ok:
            fails += 1
            bad = np.argwhere(~((g_draws == o_draws) | (np.isnan(g_draws) & np.isnan(o_draws))))
            print("MISMATCH", desc, mcmc_amd.last_kernel(), "bad chains", sorted(set(bad[:, 2].tolist()))[:6], "first", bad[:1].tolist(),
                  "gpu", g_draws[tuple(bad[0])] if len(bad) else None, "oracle", o_draws[tuple(bad[0])] if len(bad) else None, flush=True)
        elif verbose:
            print("ok      ", desc, mcmc_amd.last_kernel(), flush=True)
    return fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = sweep(n, sd)
    print("mismatching cases:", f)
    sys.exit(1 if f else 0)
