"""Randomised sweep of nuts on the LDS-streamed evaluation (mcmc_amd/csrc/nuts_lds.hpp) against literal_kernel<2> of the same library
(MI_KERNEL_LITERAL: one workgroup per chain, the reference's recursion as written, itself pinned against the oracle by
tests/test_gpu_literal_paths.py and the CPU suite) -- both run on the GPU, so the cases can be long: logistic / dense targets of every
instantiation, ragged workgroups, draw counts and adaptation windows incl. none, depth caps 1..10, step sizes from tiny to absurd, a
diagonal precond_mat, chains that start in the non-finite regime, runs cut in two.  Bit-exact or report.
Usage (GPU box): python tests/fuzz_nuts_lds.py [n_cases] [seed] [grid_cap]
grid_cap > 0: the persistent grid is capped at that many workgroups (32 chain slots each), chain counts go up to 300 and draw counts to 28 -- more chains than slots, so
chains are handed out by the counter and the runs are cut into PIECES that migrate between slots (logistic_nuts_impl.hpp)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mcmc_amd
from mcmc_amd import synth


def sweep(n_cases=30, seed=1, verbose=True, cap=0):
    rng = np.random.default_rng(seed)
    mcmc_amd.test_set_grid_cap(cap)
    fails = 0
    for case in range(n_cases):
        kind = str(rng.choice(["logistic", "dense"]))
        if kind == "logistic":
            d = int(rng.choice([9, 16, 40, 64, 65, 128, 129, 200, 256, 300, 512]))
            n_rows = int(rng.choice([1, 7, 16, 17, 50, 130]))
            X, y = synth.logistic_problem(d, n_rows, seed=int(rng.integers(1, 99)))
            tk, tkw, scale = mcmc_amd.TARGET_LOGISTIC, dict(X=X, y=y), 0.1
        else:
            d = int(rng.choice([129, 160, 192, 193, 256, 257, 384, 385, 512]))
            tk, tkw, scale = mcmc_amd.TARGET_GAUSS_DENSE, dict(prec=synth.dense_gaussian_precision(d, seed=int(rng.integers(1, 99)))), 0.5
        C = int(rng.choice([1, 5, 16, 17, 32, 33, 70, 130])) if cap == 0 else int(rng.choice([33, 70, 130, 200, 300]))
        burn, keep = (int(rng.integers(0, 10)), int(rng.integers(0, 10))) if cap == 0 else (int(rng.integers(0, 15)), int(rng.integers(0, 15)))
        if burn + keep == 0: keep = 1
        adapt = int(rng.integers(0, burn + keep + 3))
        max_depth = int(rng.choice([1, 2, 3, 5, 7, 10]))
        eps0 = float(rng.choice([0.005, 0.03, 0.1, 0.5, 2.0]))
        init = synth.initial_states(C, d, seed=int(rng.integers(1, 1000))) * scale
        wild = rng.random() < 0.25
        if wild:                                           # some chains start where energies overflow
            for c in rng.choice(C, size=min(C, 3), replace=False):
                init[c] *= float(rng.choice([1e150, 1e300])); 
                if rng.random() < 0.3: init[c, 0] = np.inf
        kw = {}
        if rng.random() < 0.3: kw["precond_mat"] = np.diag(rng.uniform(0.3, 3.0, d))
        dense_m = rng.random() < 0.25                      # round 6: a DENSE precond_mat without bounds (nuts_lds.hpp: DENSEM)
        if dense_m:
            A = rng.standard_normal((d, d)) / np.sqrt(d)
            kw["precond_mat"] = A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))
        if not dense_m and rng.random() < 0.3:             # settings.vals_bound (lds_box.hpp)
            kind_b = np.where(rng.random(d) < rng.choice([0.05, 0.5]), rng.integers(2, 5, d), 1)
            kw.update(vals_bound=1, lower_bounds=np.where((kind_b == 2) | (kind_b == 4), -1.5, -np.inf), upper_bounds=np.where((kind_b == 3) | (kind_b == 4), 2.0, np.inf))
            if not wild: init = np.clip(init, -1.0, 1.5)
        S = lambda b, k: mcmc_amd.default_settings(rng_seed_value=int(sd), n_burnin_draws=b, n_keep_draws=k, n_adapt_draws=adapt,
                                                   max_tree_depth=max_depth, step_size=eps0, **kw)
        sd = rng.integers(1, 10**6)
        chain0 = int(rng.integers(0, 5000))
        a_draws, a = mcmc_amd.sample("nuts", tk, init, S(burn, keep), chain0=chain0, want_adapt_state=True, **tkw)
        kernel = mcmc_amd.last_kernel()
        b_draws, b = mcmc_amd.sample("nuts", tk, init, S(burn, keep), chain0=chain0, want_adapt_state=True, kernel_hint=mcmc_amd.KERNEL_LITERAL, **tkw)
        bits = lambda v: np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
        same = lambda u, v: np.array_equal(bits(u), bits(v)) or np.array_equal(u, v, equal_nan=True)     # (NaN payloads may differ)
        diff = [k for k, v in dict(kernel=kernel.startswith("logit_lds_kernel<") and mcmc_amd.last_kernel().startswith("literal_kernel<"), draws=same(a_draws, b_draws),
                                   depth=np.array_equal(a["depth"], b["depth"]), n_leap=np.array_equal(a["n_leap"], b["n_leap"]), n_accept=np.array_equal(a["n_accept"], b["n_accept"]),
                                   eps=same(a["eps"], b["eps"]), theta=same(a["theta"], b["theta"]), adapt_state=same(a["adapt_state"], b["adapt_state"])).items() if not v]
        ok = not diff
        cut = None
        if ok and not wild and burn == 0 and keep >= 2 and "vals_bound" not in kw:   # (a checkpoint holds theta in the natural space: transform(inv_transform(.)) is not the identity in floating point)   # the same run cut in two on the tiled kernel (all draws kept: rows compare one to one)
            cut = int(rng.integers(1, keep))
            p_draws, p = mcmc_amd.sample("nuts", tk, init, S(0, cut), chain0=chain0, want_adapt_state=True, **tkw)
            q_draws, q = mcmc_amd.sample("nuts", tk, p["theta"].T.copy(), S(0, keep - cut), chain0=chain0, draw0=cut, step_size_in=p["eps"],
                                         adapt_state_in=p["adapt_state"], **tkw)
            ok = same(np.concatenate([p_draws, q_draws]), a_draws) and same(q["eps"], a["eps"]) and np.array_equal(p["n_leap"] + q["n_leap"], a["n_leap"])
        if verbose or not ok:
            print(("ok  " if ok else "FAIL"), dict(kind=kind, d=d, n_rows=(n_rows if kind == "logistic" else 0), C=C, burn=burn, keep=keep, adapt=adapt,
                                                   max_depth=max_depth, eps0=eps0, chain0=chain0, wild=wild, diag="precond_mat" in kw and not dense_m, dense_m=bool(dense_m), bounds="vals_bound" in kw, cut=cut,
                                                   seed=int(sd), kernel=kernel, leaps=int(a["n_leap"].sum()), differ=diff), flush=True)
        fails += 0 if ok else 1
    return fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = sweep(n, s, cap=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    print("mismatching cases:", f)
    sys.exit(1 if f else 0)
