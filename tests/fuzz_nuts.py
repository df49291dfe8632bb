"""Randomised parity sweep of the plain NUTS kernels (nuts_memo.hpp, and the tick-local nuts_async.hpp) against the recursive oracle: draw counts, burn-in / adaptation
windows, tree-depth caps (0 included), chain counts around the 16-chain wave, with and without kept draws.  Bit-exact or report.
Usage (GPU box): python tests/fuzz_nuts.py [n_cases] [seed] [grid_cap]   (test infrastructure: it drives the oracle)
grid_cap > 0: the persistent grid is capped at that many workgroups, 70-400 chains, up to 28 draws, some chains started non-finite: more chains than chain slots, so the
counter hands chains out and the runs are cut into PIECES that migrate between slots (nuts_launch.hip)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mcmc_amd
from mcmc_amd import synth
from test_gpu_parity_nuts import _oracle
import orc


def sweep(n_cases=40, seed=1, verbose=True, cap=0):
    rng = np.random.default_rng(seed)
    mcmc_amd.test_set_grid_cap(cap)
    fails = 0
    for case in range(n_cases):
        d = int(rng.choice([1, 3, 8, 16, 17, 33, 64, 100, 128]))
        C = int(rng.choice([1, 5, 16, 17, 40, 64, 70])) if cap == 0 else int(rng.choice([70, 150, 260, 400]))
        burn, keep = (int(rng.integers(0, 12)), int(rng.integers(0, 12))) if cap == 0 else (int(rng.integers(0, 15)), int(rng.integers(0, 15)))
        if burn + keep == 0: keep = 1
        adapt = int(rng.integers(0, burn + keep + 3))
        max_depth = int(rng.choice([0, 1, 2, 3, 5, 10]))
        eps0 = float(rng.choice([0.02, 0.1, 0.5, 1.0, 3.0]))
        kind = str(rng.choice(["dense", "iso", "diag"]))
        prec, kg, ko = None, mcmc_amd.TARGET_GAUSS_ISO, orc.TARGET_ISO
        if kind == "dense": prec, kg, ko = synth.dense_gaussian_precision(d, seed=int(rng.integers(1, 90))), mcmc_amd.TARGET_GAUSS_DENSE, orc.TARGET_DENSE
        elif kind == "diag": prec, kg, ko = synth.ill_conditioned_diag(d, 30.0), mcmc_amd.TARGET_GAUSS_DIAG, orc.TARGET_DIAG
        init = synth.initial_states(C, d, seed=int(rng.integers(1, 1000)))
        if cap and rng.random() < 0.3:
            for c in rng.choice(C, size=3, replace=False): init[c] *= float(rng.choice([1e150, 1e300, np.inf]))
        M = np.diag(rng.uniform(0.3, 3.0, d)) if rng.random() < 0.3 else None        # a diagonal precond_mat: nuts_gauss_memo_kernel<., true, .>
        st = mcmc_amd.default_settings(rng_seed_value=int(rng.integers(1, 10**6)), n_burnin_draws=burn, n_keep_draws=keep,
                                       n_adapt_draws=adapt, max_tree_depth=max_depth, step_size=eps0, **(dict(precond_mat=M) if M is not None else {}))
        chain0 = int(rng.integers(0, 5000))
        # AUTO / MEMO: the memoised kernel; REG: a retired kernel's hint (ignored); TICK_LOCAL: the independent tick-local kernel
        hint = int(rng.choice([mcmc_amd.KERNEL_AUTO, mcmc_amd.KERNEL_AUTO, mcmc_amd.KERNEL_NUTS_REG, mcmc_amd.KERNEL_NUTS_TICK_LOCAL, mcmc_amd.KERNEL_NUTS_MEMO_INTICK]))
        g_draws, g = mcmc_amd.nuts(kg, init, st, prec=prec, chain0=chain0, kernel_hint=hint)
        o_draws, o = _oracle(ko, d, init, st, prec=prec, chain0=chain0, **(dict(precond=M) if M is not None else {}))
        bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
        same = lambda a, b: np.array_equal(bits(a), bits(b)) or np.array_equal(a, b, equal_nan=True)     # (NaN payloads may differ)
        ok = (same(g_draws, o_draws) and np.array_equal(g["depth"], o["depth"]) and np.array_equal(g["n_leap"], o["n_leap"])
              and np.array_equal(g["n_accept"], o["n_accept"]) and same(g["eps"], o["eps"]))
        if verbose or not ok:
            print(("ok  " if ok else "FAIL"), dict(kind=kind, d=d, C=C, burn=burn, keep=keep, adapt=adapt, max_depth=max_depth, eps0=eps0, chain0=chain0, hint=hint, diag_m=M is not None, kernel=mcmc_amd.last_kernel()), flush=True)
        fails += 0 if ok else 1
    return fails


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = sweep(n, s, cap=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    print("mismatching cases:", f)
    sys.exit(1 if f else 0)
