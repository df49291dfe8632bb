"""Committed oracle regression vectors (tests/golden/oracle_kat.npz; NOT reference outputs -- the reference has none).

CPU: the live oracle still reproduces them bit for bit.  GPU: the engine reproduces them through the C ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402

KAT = np.load(os.path.join(HERE, "golden", "oracle_kat.npz"))
NAMES = [name for name, _, _ in G.cases()]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    case = {n: (t, a) for n, t, a in G.cases()}[name]
    init, out = G.run_case(*case)
    assert np.array_equal(init, KAT[f"{name}/init"])
    for k, v in out.items():
        assert np.array_equal(v, KAT[f"{name}/{k}"]), (name, k)


def test_golden_vectors_are_sane():
    for name in NAMES:
        dr = KAT[f"{name}/draws"]                       # [C, n_keep, d]
        assert dr.shape[:2] == (G.C, G.KEEP) and np.isfinite(dr).all()
        acc = KAT[f"{name}/accept"]
        assert acc.shape == (G.C, G.BURN + G.KEEP) and set(np.unique(acc)) <= {0, 1}
    assert KAT["nuts_dense8/depth"].max() <= 10 and KAT["nuts_dense8/depth"].min() >= 1


GPU_CASES = list(NAMES)      # nuts_logit5 runs on the one-lane-per-chain engine (LogisticSmallModel), the rest as before


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_CASES)
def test_engine_reproduces_golden(name):
    import mcmc_amd
    algo, tname = name.split("_")
    t, a = {n: (t, a) for n, t, a in G.cases()}[name]
    kind = {"iso3": mcmc_amd.TARGET_GAUSS_ISO, "dense8": mcmc_amd.TARGET_GAUSS_DENSE, "logit5": mcmc_amd.TARGET_LOGISTIC,
            "normal2": mcmc_amd.TARGET_NORMAL_MODEL, "logit40": mcmc_amd.TARGET_LOGISTIC, "dense160": mcmc_amd.TARGET_GAUSS_DENSE,
            "logit40box": mcmc_amd.TARGET_LOGISTIC, "dense160box": mcmc_amd.TARGET_GAUSS_DENSE}[tname]
    bkw = dict(vals_bound=1, lower_bounds=t["lower"], upper_bounds=t["upper"]) if "lower" in t else {}
    st = mcmc_amd.default_settings(rng_seed_value=G.SEED, n_burnin_draws=G.BURN, n_keep_draws=G.KEEP,
                                   n_leap_steps=a.get("n_leap", 1), step_size=a["step"], n_adapt_draws=a.get("n_adapt", 1000),
                                   n_fp_steps=a.get("n_fp", 5), max_tree_depth=a.get("max_depth", 10), **bkw)
    draws, g = mcmc_amd.sample(algo, kind, KAT[f"{name}/init"], st, prec=t.get("prec"), X=t.get("X"), y=t.get("y"))
    want = np.transpose(KAT[f"{name}/draws"], (1, 2, 0))            # [C, n_keep, d] -> [n_keep, d, C]
    assert np.array_equal(draws, want)
    if tname in ("logit40", "dense160", "logit40box", "dense160box"): assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    assert np.array_equal(g["n_accept"], KAT[f"{name}/accept"][:, G.BURN:].sum(axis=1).astype(np.uint64))
    if algo == "nuts":
        assert np.array_equal(g["depth"], KAT[f"{name}/depth"].T)
        assert np.array_equal(g["eps"], KAT[f"{name}/eps"])
