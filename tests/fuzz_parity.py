"""Randomised parity sweep: every device sampler against the oracle on random small cases (bit-exact or report).
Usage (GPU box): python tests/fuzz_parity.py [n_cases] [seed]   (test infrastructure: it drives the oracle)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mcmc_amd, orc
from mcmc_amd import synth



def sweep(n_cases=60, seed=1, verbose=True):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    fails = 0
    say = print if verbose else (lambda *a, **k: None)
    def bounds(d):
        kind = rng.integers(1, 5, d)
        if rng.random() < 0.5:                 # sparse bounds: most 4-dim slices have no bounded dimension (the kernels skip those)
            kind = np.where(rng.random(d) < 0.12, kind, 1)
        lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf)
        ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
        return lb, ub


    for case in range(n_cases):
        algo = ["hmc", "mala", "nuts", "rwmh"][case % 4]
        tgt = rng.choice(["dense", "iso", "diag", "logit"] if algo in ("hmc", "mala", "rwmh") else ["dense", "iso", "diag"])
        d = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 33, 64, 100, 128])) if tgt != "logit" else int(rng.choice([3, 17, 64, 65, 130, 300]))
        if tgt == "dense" and rng.random() < 0.3: d = int(rng.choice([129, 192, 200, 256, 300, 384, 385, 512]))    # P streamed through LDS (hmc / mala / rwmh), literal (nuts, bounds, dense M)
        C = int(rng.choice([1, 3, 16, 17, 33, 70]))
        rseed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
        burn, keep = int(rng.integers(0, 4)), int(rng.integers(1, 7))
        eps = float(rng.choice([0.01, 0.05, 0.2, 0.7, 1.5]))
        L = int(rng.integers(0, 6))
        general = rng.random() < (0.4 if tgt != "logit" else 0.3)
        if d > 128 and algo == "mala": general = general and rng.random() < 0.5
        kw, okw = {}, {}
        prec = X = y = None
        if tgt == "dense": prec, kg, ko = synth.dense_gaussian_precision(d, seed=rseed % 97), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
        elif tgt == "diag": prec, kg, ko = synth.ill_conditioned_diag(d, 20.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
        elif tgt == "iso": kg, ko = mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
        else:
            N = int(rng.choice([1, 15, 16, 40, 100])); X, y = synth.logistic_problem(d, N, seed=rseed % 89); kg, ko = mcmc_amd.TARGET_LOGISTIC, orc.TARGET_LOGISTIC
        scale = float(rng.choice([0.1, 1.0, 3.0]))
        init = synth.initial_states(C, d, seed=rseed % 1013) * scale
        if algo in ("hmc", "mala", "nuts") and rng.random() < 0.15:      # the non-finite regime: chains that blow up or start non-finite
            eps = float(rng.choice([30.0, 1.0e5, 1.0e160]))
            if rng.random() < 0.5: init[int(rng.integers(0, C)), int(rng.integers(0, d))] = float(rng.choice([np.inf, -np.inf, np.nan]))
        big = d > 128
        if big and general: C = min(C, 3)                  # (the literal kernels and the CPU oracle are O(d^2) .. O(d^3) per step and chain)
        if general:
            if rng.random() < (0.7 if not (big and algo == "mala") and tgt != "logit" else 0.25) and not (big and algo == "mala"):
                lb, ub = bounds(d); kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
                init = np.clip(init, -1.0, 1.5)
            if rng.random() < 0.6 or not kw:
                M = np.diag(rng.uniform(0.3, 3.0, d))
                # dense precond / cov (d <= 64; mala: unbounded; hmc without bounds: any d -- beyond d = 128 and on the logistic target the
                # LDS-streamed kernel streams INV(M) and CHOL_LOWER(M) too, logistic_lds.hpp DENSEM)
                if (algo != "mala" or "vals_bound" not in kw) and (d <= 64 or (algo == "hmc" and "vals_bound" not in kw)) and rng.random() < 0.4:
                    A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + M
                kw.update(precond_mat=M); okw.update(precond=M)
        tkw = {}
        if tgt == "dense" and d > 128:                      # dot products over the streamed kernel's four dimension quarters
            dq = 48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128
            tkw = dict(blocks=4, block_size=dq); okw.update(blocks=4, block_size=dq)
        if tgt == "logit":
            dq = 16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128
            tkw = dict(blocks=4, block_size=dq, eta_chains=2); okw.update(blocks=4, block_size=dq)
        st = mcmc_amd.default_settings(rng_seed_value=rseed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps,
                                       n_adapt_draws=burn, max_tree_depth=int(rng.integers(0, 7)), **kw)
        t = orc.TargetSpec(ko, d, prec=prec, X=X, y=y, W=4, **tkw)
        s = orc.make_settings(seed=rseed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, n_adapt=burn, max_depth=int(st.max_tree_depth), W=4, hoist=1, **okw)
        desc = f"{algo} {tgt} d={d} C={C} eps={eps} L={L} burn={burn} keep={keep} general={sorted(kw)} depth={int(st.max_tree_depth)}"
        try:
            g_draws, g = mcmc_amd.sample(algo, kg, init, st, prec=prec, X=X, y=y, chain0=chain0)
        except mcmc_amd.MiMcmcError as e:
            say("REFUSED", desc, "->", str(e)[:90]); continue
        o_draws, o = orc.run_many({"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS, "rwmh": orc.ALGO_RWMH}[algo], t, init, s, chain0=chain0)
        ok = np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["n_accept"], o["n_accept"])
        if algo == "nuts": ok = ok and np.array_equal(g["n_leap"], o["n_leap"]) and np.array_equal(g["eps"], o["eps"], equal_nan=True)
        if not ok:
            fails += 1
            if os.environ.get("MI_FUZZ_DUMP"):            # the case's inputs and both results, for a post-mortem off the box
                np.savez(os.path.join(os.environ["MI_FUZZ_DUMP"], f"fuzz_mismatch_{seed}_{case}.npz"), algo=algo, tgt=tgt, d=d, C=C, eps=eps, L=L, burn=burn, keep=keep,
                         rseed=rseed, chain0=chain0, init=init, prec=(prec if prec is not None else np.zeros(0)), M=kw.get("precond_mat", np.zeros(0)),
                         lb=kw.get("lower_bounds", np.zeros(0)), ub=kw.get("upper_bounds", np.zeros(0)), g_draws=g_draws, o_draws=o_draws,
                         g_acc=g["n_accept"], o_acc=o["n_accept"], kernel=mcmc_amd.last_kernel())
            bad = np.argwhere(~np.isclose(g_draws, o_draws, rtol=0, atol=0, equal_nan=True))
            fields = [k for k in ("n_accept", "n_leap", "eps") if k in o and not np.array_equal(g[k], o[k], equal_nan=True)]
            say("MISMATCH", desc, "first bad index", bad[:1].tolist(), "fields", fields,
                  {k: [(int(i), float(g[k][i]), float(o[k][i])) for i in np.flatnonzero(~((g[k] == o[k]) | (np.isnan(g[k].astype(float)) & np.isnan(o[k].astype(float)))))[:4]] for k in fields},
                  'n_leap', [(int(g['n_leap'][i]), int(o['n_leap'][i])) for i in np.flatnonzero(g['eps'] != o['eps'])[:4]] if 'eps' in fields else '')
        else:
            say("ok      ", desc)
    return fails


def sweep_dense_m(n_cases=40, seed=1, verbose=True):
    """hmc and mala with a DENSE precond_mat on the LDS-streamed kernels (logistic_lds.hpp DENSEM, round 5): dense Gaussians with 128 < d <= 512 and the
    logistic target with 8 < d <= 512 -- every instantiation, ragged workgroups, 0..5 leapfrog steps, step sizes up to the non-finite regime
    and non-finite starts (flagged, replayed literally with the same matrices), continuation offsets.  Returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    fails = 0
    say = print if verbose else (lambda *a, **k: None)
    for case in range(n_cases):
        tgt = "dense" if case % 2 == 0 else "logit"
        algo = "hmc" if (case // 2) % 2 == 0 else "mala"
        d = int(rng.choice([129, 144, 192, 193, 250, 256, 257, 384, 385, 512])) if tgt == "dense" else int(rng.choice([9, 16, 17, 64, 65, 128, 129, 256, 300, 512]))
        C = int(rng.choice([1, 3, 16, 17, 33, 70]))
        if d > 256: C = min(C, 17)                       # (the oracle inverts M once per chain)
        rseed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
        burn, keep, L = int(rng.integers(0, 3)), int(rng.integers(1, 5)), int(rng.integers(0, 6))
        eps = float(rng.choice([0.01, 0.05, 0.2, 0.5]))
        A = rng.standard_normal((d, d)) / np.sqrt(d); M = A @ A.T + np.diag(rng.uniform(0.3, 3.0, d))
        init = synth.initial_states(C, d, seed=rseed % 1013) * float(rng.choice([0.1, 0.5, 1.0]))
        if rng.random() < 0.25:
            eps = float(rng.choice([30.0, 1.0e5, 1.0e160]))
            if rng.random() < 0.6: init[int(rng.integers(0, C)), int(rng.integers(0, d))] = float(rng.choice([np.inf, -np.inf, np.nan, 1e300]))
        prec = X = y = None
        if tgt == "dense":
            prec, kg, ko = synth.dense_gaussian_precision(d, seed=rseed % 97), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
            blk = dict(blocks=4, block_size=48 if d <= 192 else 64 if d <= 256 else 96 if d <= 384 else 128); tkw = dict(blk)
        else:
            N = int(rng.choice([1, 15, 16, 17, 40, 100])); X, y = synth.logistic_problem(d, N, seed=rseed % 89); kg, ko = mcmc_amd.TARGET_LOGISTIC, orc.TARGET_LOGISTIC
            blk = dict(blocks=4, block_size=16 if d <= 64 else 32 if d <= 128 else 64 if d <= 256 else 128); tkw = dict(blk, eta_chains=2)
        st = mcmc_amd.default_settings(rng_seed_value=rseed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps, precond_mat=M)
        s = orc.make_settings(seed=rseed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1, precond=M, **blk)
        desc = f"{algo} {tgt} dense precond d={d} C={C} eps={eps} L={L} burn={burn} keep={keep}"
        g_draws, g = mcmc_amd.sample(algo, kg, init, st, prec=prec, X=X, y=y, chain0=chain0)
        kern = mcmc_amd.last_kernel()
        o_draws, o = orc.run_many(orc.ALGO_HMC if algo == "hmc" else orc.ALGO_MALA, orc.TargetSpec(ko, d, prec=prec, X=X, y=y, W=4, **tkw), init, s, chain0=chain0)
        ok = (kern.startswith("logit_lds_kernel<") and kern.endswith("false, false, true>") and np.array_equal(g_draws, o_draws, equal_nan=True)
              and np.array_equal(g["n_accept"], o["n_accept"]) and np.array_equal(g["n_leap"], o["n_leap"]))
        if not ok:
            fails += 1
            bad = np.argwhere(~((g_draws == o_draws) | (np.isnan(g_draws) & np.isnan(o_draws))))
            say("MISMATCH", desc, kern, "first bad index", bad[:1].tolist())
        else:
            say("ok      ", desc, "nan" if np.isnan(o_draws).any() else "")
    return fails


def sweep_small(n_cases=60, seed=1, verbose=True):
    """The one-lane-per-chain engine (hmc / mala / rwmh / rmhmc / nuts on the d = 2 normal model): random observations, bounds of every
    type, diagonal / dense preconditioners, degenerate sizes, step sizes that blow the chain up.  Returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    fails = 0
    say = print if verbose else (lambda *a, **k: None)
    algos = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "rwmh": orc.ALGO_RWMH, "rmhmc": orc.ALGO_RMHMC, "nuts": orc.ALGO_NUTS}
    for case in range(n_cases):
        algo = ["hmc", "mala", "rwmh", "rmhmc", "nuts"][case % 5]
        n = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 100, 1000]))
        C = int(rng.choice([1, 3, 63, 64, 65, 130]))
        rseed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
        burn, keep = int(rng.integers(0, 4)), int(rng.integers(1, 8))
        eps = float(rng.choice([0.005, 0.02, 0.08, 0.3, 1.0, 4.0]))
        L, n_fp = int(rng.integers(0, 5)), int(rng.integers(0, 6))
        x = float(rng.uniform(-3, 3)) + float(rng.uniform(0.2, 3.0)) * rng.standard_normal(n)
        init = np.stack([x.mean() + rng.uniform(-1.0, 1.0, C), np.abs(x.std()) + rng.uniform(0.2, 2.0, C)], axis=1)
        if rng.random() < 0.1: init[:, 1] = -init[:, 1]                      # sigma < 0: log of a negative number from the start
        kw, okw = {}, {}
        if rng.random() < 0.5:
            kind = rng.integers(1, 5, 2)
            lo, hi = init.min(axis=0) - rng.uniform(0.1, 2.0, 2), init.max(axis=0) + rng.uniform(0.1, 2.0, 2)
            lb = np.where((kind == 2) | (kind == 4), lo, -np.inf); ub = np.where((kind == 3) | (kind == 4), hi, np.inf)
            kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
        if algo != "rmhmc" and rng.random() < 0.6:
            M = np.diag(rng.uniform(0.3, 3.0, 2))
            if rng.random() < 0.6:
                A = rng.standard_normal((2, 2)); M = A @ A.T / 2 + M
            kw.update(precond_mat=M); okw.update(precond=M)
        n_adapt, depth = int(rng.integers(0, burn + keep + 2)), int(rng.integers(0, 8))
        st = mcmc_amd.default_settings(rng_seed_value=rseed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps,
                                       n_fp_steps=n_fp, n_adapt_draws=n_adapt, max_tree_depth=depth, **kw)
        t = orc.TargetSpec(orc.TARGET_NORMAL_MODEL, 2, y=x, W=1)
        s = orc.make_settings(seed=rseed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, n_fp=n_fp, W=1, hoist=int(rng.integers(0, 2)),
                              n_adapt=n_adapt, max_depth=depth, **okw)
        desc = f"{algo} normal-model n={n} C={C} eps={eps} L={L} n_fp={n_fp} burn={burn} keep={keep} adapt={n_adapt} depth={depth} general={sorted(kw)}"
        g_draws, g = mcmc_amd.sample(algo, mcmc_amd.TARGET_NORMAL_MODEL, init, st, y=x, chain0=chain0)
        o_draws, o = orc.run_many(algos[algo], t, init, s, chain0=chain0)
        ok = (np.array_equal(g_draws, o_draws, equal_nan=True) and np.array_equal(g["n_accept"], o["n_accept"])
              and np.array_equal(g["n_leap"], o["n_leap"]))
        if algo == "nuts": ok = ok and np.array_equal(g["eps"], o["eps"], equal_nan=True)
        if not ok:
            fails += 1
            bad = np.argwhere(~((g_draws == o_draws) | (np.isnan(g_draws) & np.isnan(o_draws))))
            say("MISMATCH", desc, "first bad index", bad[:1].tolist(), "nan in oracle", bool(np.isnan(o_draws).any()))
        else:
            say("ok      ", desc, "nan" if np.isnan(o_draws).any() else "")
    return fails


def sweep_mass(n_cases=40, seed=1, verbose=True):
    """hmc with PER-CHAIN diagonal masses (mi_chains.mass_diag): chain c against the oracle run with precond_mat = diag(mass[:, c]) --
    elementwise kernels (iso / diag targets, any d), literal kernels (dense, logistic, bounds), the non-finite regime.  Returns the mismatches."""
    rng = np.random.default_rng(seed)
    fails = 0
    say = print if verbose else (lambda *a, **k: None)
    for case in range(n_cases):
        tgt = rng.choice(["iso", "diag", "dense", "logit"])
        d = int(rng.choice([1, 2, 5, 17, 64, 129, 300])) if tgt in ("iso", "diag") else int(rng.choice([2, 9, 33, 70]))
        C = int(rng.choice([1, 3, 17, 70])) if tgt in ("iso", "diag") else int(rng.choice([1, 3, 5]))
        rseed = int(rng.integers(1, 10**6)); chain0 = int(rng.integers(0, 1000))
        burn, keep, L = int(rng.integers(0, 3)), int(rng.integers(1, 6)), int(rng.integers(0, 5))
        eps = float(rng.choice([0.02, 0.1, 0.4, 30.0, 1e160]))
        prec = X = y = None; tkw = {}; okw = {}; kw = {}
        if tgt == "dense": prec, kg, ko = synth.dense_gaussian_precision(d, seed=rseed % 97), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
        elif tgt == "diag": prec, kg, ko = synth.ill_conditioned_diag(d, 20.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
        elif tgt == "iso": kg, ko = mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
        else:
            X, y = synth.logistic_problem(d, int(rng.choice([1, 16, 40])), seed=rseed % 89); kg, ko = mcmc_amd.TARGET_LOGISTIC, orc.TARGET_LOGISTIC
            dq = 16 if d <= 64 else 32
            tkw = dict(blocks=4, block_size=dq, eta_chains=2); okw.update(blocks=4, block_size=dq)
        init = synth.initial_states(C, d, seed=rseed % 1013) * float(rng.choice([0.1, 1.0]))
        if rng.random() < 0.2: init[int(rng.integers(0, C)), int(rng.integers(0, d))] = float(rng.choice([np.inf, np.nan, 1e300]))
        if tgt != "logit" and rng.random() < 0.3:
            kind = rng.integers(1, 5, d)
            lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
            kw.update(vals_bound=1, lower_bounds=lb, upper_bounds=ub); okw.update(lower=lb, upper=ub)
            init = np.where(np.isfinite(init), np.clip(init, -1.0, 1.5), init)
        mass = np.ascontiguousarray(rng.uniform(0.2, 5.0, (d, C)))
        st = mcmc_amd.default_settings(rng_seed_value=rseed, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps, **kw)
        t_g = mcmc_amd.make_target(kg, d, prec=prec, X=X, y=y)
        theta = np.ascontiguousarray(init.T.copy()); draws = np.zeros((keep, d, C)); nacc = np.zeros(C, dtype=np.uint64)
        mcmc_amd.run("hmc", t_g, st, mcmc_amd.make_chains(theta, C, chain0=chain0, draws=draws, n_accept=nacc, mass_diag=mass))
        t = orc.TargetSpec(ko, d, prec=prec, X=X, y=y, W=4, **tkw)
        ok = True
        for c in range(C):
            s = orc.make_settings(seed=rseed, n_burnin=burn, n_keep=keep, n_leap=L, step=eps, W=4, hoist=1, precond=np.diag(mass[:, c]), **okw)
            o, info = orc.run_many(orc.ALGO_HMC, t, init[c:c + 1], s, chain0=chain0 + c)
            ok = ok and np.array_equal(draws[:, :, c], o[:, :, 0], equal_nan=True) and int(nacc[c]) == int(info["n_accept"][0])
        desc = f"hmc per-chain mass {tgt} d={d} C={C} eps={eps} L={L} burn={burn} keep={keep} general={sorted(kw)} kernel={mcmc_amd.last_kernel()}"
        if not ok: fails += 1
        say("ok      " if ok else "MISMATCH", desc)
    return fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "mass":
        f = sweep_mass(n, sd)
        print(f"{n} per-chain-mass cases, {f} mismatches")
        sys.exit(1 if f else 0)
    f = sweep(n, sd) + sweep_small(n, sd)
    print(f"2 x {n} cases, {f} mismatches")
    sys.exit(1 if f else 0)
