"""Generates tests/golden/oracle_kat.npz -- ORACLE regression vectors, NOT reference outputs.

kthohr/mcmc ships no golden vectors and cannot be built in this image, so nothing here
comes from the reference.  These vectors freeze the CPU oracle's own outputs on the known-answer shapes SURVEY.md
8(c) lists (HMC / MALA / NUTS / RWMH x {iso Gaussian d=3, dense Gaussian d=8, logistic d=5}, and RM-HMC on the d=2 normal
model; n_burnin=5, n_keep=20, fixed seeds), so that a later change of the oracle, of its math, or of the RNG layout is caught on the CPU, and so that
the GPU tests have a second, committed reference next to the live oracle.

    python tests/golden/make_golden.py        # rewrites oracle_kat.npz (only when the definition changes on purpose)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from mcmc_amd import synth  # noqa: E402

SEED, BURN, KEEP, C = 20240926, 5, 20, 4


def cases():
    P8 = synth.dense_gaussian_precision(8, seed=5)
    X5, y5 = synth.logistic_problem(5, 40, seed=4)
    targets = {
        "iso3": dict(kind=orc.TARGET_ISO, d=3),
        "dense8": dict(kind=orc.TARGET_DENSE, d=8, prec=P8),
        "logit5": dict(kind=orc.TARGET_LOGISTIC, d=5, X=X5, y=y5, blocks=4, block_size=16, eta_chains=2),
    }
    algos = {
        "hmc": dict(algo=orc.ALGO_HMC, n_leap=5, step=0.2),
        "mala": dict(algo=orc.ALGO_MALA, step=0.15),
        "nuts": dict(algo=orc.ALGO_NUTS, step=1.0, n_adapt=10),
        "rwmh": dict(algo=orc.ALGO_RWMH, step=0.4),
    }
    for tn, t in targets.items():
        for an, a in algos.items():
            yield f"{an}_{tn}", t, a
    # mcmc::rmhmc on the d = 2 normal model of the reference's example (examples/eigen/rmhmc_normal.cpp), Fisher metric
    x = 2.0 + 2.0 * synth.initial_states(60, 1, seed=17)[:, 0]
    normal2 = dict(kind=orc.TARGET_NORMAL_MODEL, d=2, y=x, init_shift=np.array([2.0, 2.5]), W=1)
    yield "rmhmc_normal2", normal2, dict(algo=orc.ALGO_RMHMC, n_leap=2, step=0.03, n_fp=5)
    # the other samplers on the same model (examples/eigen/{hmc,mala,nuts}_normal.cpp), one-lane-per-chain engine: sequential dots
    yield "hmc_normal2", normal2, dict(algo=orc.ALGO_HMC, n_leap=3, step=0.1)
    yield "mala_normal2", normal2, dict(algo=orc.ALGO_MALA, step=0.15)
    yield "nuts_normal2", normal2, dict(algo=orc.ALGO_NUTS, step=1.0, n_adapt=10)
    yield "rwmh_normal2", normal2, dict(algo=orc.ALGO_RWMH, step=0.15)
    # round 4: the LDS-streamed evaluation under nuts (nuts_lds.hpp) and with bounds (lds_box.hpp) -- reduction orders over four dimension quarters
    X40, y40 = synth.logistic_problem(40, 30, seed=6)
    P160 = synth.dense_gaussian_precision(160, seed=7)
    logit40 = dict(kind=orc.TARGET_LOGISTIC, d=40, X=X40, y=y40, blocks=4, block_size=16, eta_chains=2, init_scale=0.2)
    dense160 = dict(kind=orc.TARGET_DENSE, d=160, prec=P160, blocks=4, block_size=48)
    box = lambda d: dict(lower=np.where(np.arange(d) % 3 == 0, -1.5, -np.inf), upper=np.where(np.arange(d) % 4 == 0, 2.0, np.inf))
    yield "nuts_logit40", logit40, dict(algo=orc.ALGO_NUTS, step=0.1, n_adapt=10, max_depth=6)
    yield "nuts_dense160", dense160, dict(algo=orc.ALGO_NUTS, step=0.1, n_adapt=10, max_depth=6)
    yield "hmc_logit40box", dict(logit40, **box(40)), dict(algo=orc.ALGO_HMC, n_leap=4, step=0.05)
    yield "nuts_dense160box", dict(dense160, **box(160)), dict(algo=orc.ALGO_NUTS, step=0.05, n_adapt=10, max_depth=5)


def run_case(t, a):
    d = t["d"]
    blocks, bs = t.get("blocks", 0), t.get("block_size", 0)
    W = t.get("W", 4)
    tgt = orc.TargetSpec(t["kind"], d, prec=t.get("prec"), X=t.get("X"), y=t.get("y"), W=W, blocks=blocks, block_size=bs,
                         eta_chains=t.get("eta_chains", 1))
    init = synth.initial_states(C, d, seed=99) * t.get("init_scale", 0.5) + t.get("init_shift", 0.0)
    if "lower" in t: init = np.clip(init, -1.0, 1.5)
    out = dict(draws=[], accept=[], depth=[], eps=[], n_leap=[])
    for c in range(C):
        s = orc.make_settings(seed=SEED, n_burnin=BURN, n_keep=KEEP, n_leap=a.get("n_leap", 1), step=a["step"],
                              n_adapt=a.get("n_adapt", 1000), W=W, hoist=1, blocks=blocks, block_size=bs, chain_id=c,
                              n_fp=a.get("n_fp", 5), max_depth=a.get("max_depth", 10), lower=t.get("lower"), upper=t.get("upper"))
        dr, info = orc.run_chain(a["algo"], tgt, init[c], s, traces=True)
        out["draws"].append(dr); out["accept"].append(info["accept"]); out["depth"].append(info["depth"])
        out["eps"].append(info["eps"]); out["n_leap"].append(info["n_leap"])
    return init, {k: np.array(v) for k, v in out.items()}


def main():
    blob = {}
    for name, t, a in cases():
        init, out = run_case(t, a)
        blob[f"{name}/init"] = init
        for k, v in out.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
