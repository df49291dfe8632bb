#!/usr/bin/env python3
"""bench.py -- throughput of the many-chain sampling hot path on MI355X, one JSON line per run.

Headline workload = BASELINE.json configs[1] (SURVEY.md 8(d) "C2"): mcmc::hmc on a d=128 correlated Gaussian
(P = A A^T / d + I, analytic gradient), 65 536 chains, fp64, step_size 0.05, n_leap_steps 16, 100 burn-in + 100 kept
draws.  `--config 2|3|4|5` runs ONE of the single-GPU BASELINE configs (MALA d=512 logistic regression, 262 144 chains;
NUTS d=128, 65 536 chains, depth 10; one GPU's 131 072-chain shard of the d=1024 ill-conditioned HMC run), each with its
own roofline object.  One "step" = one mi_mcmc_<algo>_run call = that whole sampling run for every chain of the rank, with
target, initial states and output buffers already resident in HBM.

Multi-GPU (one rank per GPU: under torchrun, or `--gpus N` alone -- bench.py then launches the N ranks itself through
torch.distributed.run and FAILS if fewer than N GPUs are visible): chains shard by global chain id (mcmc_amd.dist.shard_bounds) with no data-path
collective.  north_star asks for STRONG scaling at 65 536 chains, so with WORLD_SIZE > 1 the total chain count stays
fixed and every rank takes its shard ("scaling": "strong"); `--scaling weak` keeps the per-GPU count fixed instead.
With more than one rank the line ALWAYS carries the one exchange the path has -- the collation of draws_out, through the C ABI
(mi_mcmc_allgather_draws_rank_major, then _begin / _wait in 4 chunks under the sampling): `value` (sampling only), `value_incl_collation`
(sampling, then one blocking all-gather), `value_overlapped`, `collate_GBps`, and the RCCL rank count.  `--collate` does the same at one GPU
(a one-rank communicator: the code path, not a transfer).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (named by the engine: mi_mcmc_last_kernel) against the resource that bounds it: algorithmic
                  flops per launch / HIP-event duration measured here on the launch stream; `traffic` = HBM bytes per
                  launch, measured by two rocprofv3 --pmc child runs of this workload when rocprofv3 is on the box (--traffic),
                  else from the committed passes of this exact workload (profiles/r*_c<N>_pmc.json), else null
  other_configs-- (default run, one GPU) BASELINE configs 3, 4, 5 under the same clock at --other-steps steps, each with its roofline, its own
                  cpu_baseline and -- where BASELINE's frozen settings do not converge (configs[2], configs[4]) -- a `converged` leg outside the
                  timed region that makes ESS/sec an estimate (CONVERGED below, frozen in BASELINE.md section 9)
  scaling_proxy-- (default run, one GPU) the share of one GPU at N = 1, 2, 4, 8 (chains_total / N) of configs[1] and configs[3], timed here:
                  a one-GPU PROXY of north_star's strong-scaling curve, labelled as such
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm) timed on this box's host cores on a
                  bounded sample of the same workload: Mode A (reference-faithful work profile) and Mode B
                  (optimised CPU: same bits, gradient reuse, no identity mat-vecs, SIMD mat-vec), built with the
                  reference's own flags (-O3 -march=native -ffp-contract=fast -fopenmp, ref: configure:196-214).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix (= vector) peak; the in-image guide lists no fp64 figure

# d, chains (per GPU when weak / total when strong), sampler settings: SURVEY.md 8(d), frozen in BASELINE.md
WORKLOADS = {
    2: dict(algo="hmc", d=128, chains=65536, n_leap_steps=16, step_size=0.05, n_burnin_draws=100, n_keep_draws=100, seed=2024,
            name="BASELINE configs[1]: mcmc::hmc, d=128 dense-precision Gaussian (P=AA^T/d+I), analytic grad, fp64",
            metric="leapfrog-steps/sec (chains*dims*steps/s), HMC d=128 correlated Gaussian, 65536 chains",
            unit="chain*dim*leapfrog-steps/s", kernel="hmc_gauss_mfma_kernel<8, 8, false, false, false>", bound="mfma"),
    3: dict(algo="mala", d=512, n_rows=1024, chains=262144, step_size=0.02, n_burnin_draws=100, n_keep_draws=100, seed=6,
            name="BASELINE configs[2]: mcmc::mala, d=512 Bayesian logistic regression (N=1024 synthetic rows), fp64",
            metric="MALA draws/sec (chains*dims*draws/s), d=512 logistic regression, 262144 chains",
            unit="chain*dim*draws/s", kernel="logit_lds_kernel<8, 0, 0, false>", bound="mfma"),
    4: dict(algo="nuts", d=128, chains=65536, n_burnin_draws=100, n_keep_draws=100, n_adapt_draws=100, max_tree_depth=10, seed=2024,
            name="BASELINE configs[3]: mcmc::nuts, d=128 dense-precision Gaussian, max_tree_depth=10, dual averaging, fp64",
            metric="leapfrog-steps/sec (chains*dims*executed steps/s), NUTS d=128 Gaussian, 65536 chains",
            unit="chain*dim*leapfrog-steps/s", kernel="nuts_gauss_memo_kernel<8, false, true>", bound="mfma"),
    5: dict(algo="hmc", d=1024, chains=131072, n_leap_steps=32, step_size=0.005, n_burnin_draws=20, n_keep_draws=8, seed=8,
            name="BASELINE configs[4], one GPU's shard: mcmc::hmc, d=1024 diagonal Gaussian (cond 1e4), 131072 of 2^20 chains, fp64",
            metric="leapfrog-steps/sec (chains*dims*steps/s), HMC d=1024 ill-conditioned Gaussian, 131072 chains per GPU",
            unit="chain*dim*leapfrog-steps/s", kernel="hmc_diag4_kernel", bound="valu-fp64"),
}


def flop_per_unit(cfg):
    """Algorithmic fp64 flops per metric unit."""
    if cfg["algo"] == "mala":
        return 4 * cfg["n_rows"]            # two N x d products (eta = X beta, X^T r) per evaluation, one evaluation per draw
    if cfg["d"] <= 128:
        return 2 * cfg["d"] + 8             # dense mat-vec with end-of-step gradient reuse + two half-kicks + drift
    return 10                               # diagonal target: lambda*theta, kick (3), kick (3), drift (2) + the shared half-step


def usable_cores():
    """Threads the host really grants: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def build_inputs(cfg, C, chain0):
    """Host-side synthetic inputs of a workload for chains [chain0, chain0 + C): (target kwargs, init [C, d])."""
    from mcmc_amd import synth
    d = cfg["d"]
    if cfg["algo"] == "mala":
        X, y = synth.logistic_problem(d, cfg["n_rows"])
        init = np.zeros((C, d))
        init[:, 0] = -0.5 + (chain0 + np.arange(C)) / float(max(1, cfg["chains"] - 1))       # distinct starts, by global chain id
        return dict(X=X, y=y), init
    if d > 128:
        prec = synth.ill_conditioned_diag(d, 1.0e4)
        return dict(prec=prec), synth.initial_states(C, d, seed=3, chain0=chain0) / np.sqrt(prec)[None, :]
    return dict(prec=synth.dense_gaussian_precision(d)), synth.initial_states(C, d, seed=3, chain0=chain0)


def cpu_baseline(cfg_id, cfg):
    """The oracle on the host cores, bounded sample of the same workload; Mode A and (hmc) Mode B."""
    cores = usable_cores()
    flags = "-O3 -march=native -ffp-contract=fast -DNDEBUG -fopenmp (ref: configure:196-214), gcc"
    fast = os.path.join(ROOT, "oracle", "liboracle_fast.so")
    try:        # -march=native: built on the box it runs on
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["ORC_LIB"] = fast
    except (OSError, subprocess.CalledProcessError):
        flags = "-O3 -mavx2 -mfma -ffp-contract=off -fopenmp (the parity build: no compiler on this box for the -march=native build)"
    import orc   # tests/orc.py: ctypes binding of oracle/liboracle*.so
    d = cfg["d"]
    n_tot = cfg["n_burnin_draws"] + cfg["n_keep_draws"]
    # chains per core so that each mode is ~5-15 s of wall time on the box's cores (measured per-chain costs)
    per_core = {2: (64, 1024), 3: (1, None), 4: (16, 24), 5: (6, 2048)}[cfg_id]
    out = {"unit": cfg["unit"], "cores": cores, "kind": "port", "flags": flags}
    for mode, tag in ((0, "mode_a"), (1, "mode_b")):
        if per_core[mode] is None:
            continue
        n_chains = per_core[mode] * cores
        kw, init = build_inputs(cfg, n_chains, 0)
        kind = {"hmc": orc.TARGET_DENSE if d <= 128 else orc.TARGET_DIAG, "nuts": orc.TARGET_DENSE, "mala": orc.TARGET_LOGISTIC}[cfg["algo"]]
        tgt = orc.TargetSpec(kind, d, W=1, **kw)                       # reference-shaped reduction order
        st = orc.make_settings(seed=cfg["seed"], n_burnin=cfg["n_burnin_draws"], n_keep=cfg["n_keep_draws"],
                               n_leap=cfg.get("n_leap_steps", 1), step=cfg.get("step_size", 1.0),
                               n_adapt=cfg.get("n_adapt_draws", 1000), max_depth=cfg.get("max_tree_depth", 10),
                               W=1, hoist=0, work_mode=mode)
        algo = {"hmc": orc.ALGO_HMC, "mala": orc.ALGO_MALA, "nuts": orc.ALGO_NUTS}[cfg["algo"]]
        if cfg["algo"] == "nuts" and mode == 1:      # the optimised CPU for nuts: the same memoised evaluation the device kernel runs (orc_nuts_memo)
            algo, st.work_mode = orc.ALGO_NUTS_MEMO, 0
        t0 = time.perf_counter()
        _, info = orc.run_many(algo, tgt, init, st, n_threads=cores, want_draws=False)
        dt = time.perf_counter() - t0
        units = float(n_chains) * d * n_tot if cfg["algo"] == "mala" else float(info["n_leap"].sum()) * d
        out[tag] = {"value": units / dt, "chains": n_chains, "wall_s": dt}
    out["value"] = out["mode_a"]["value"]                               # the reference's work profile is THE baseline
    out["mode_a"]["what"] = ("reference-faithful work profile: every callback of the reference (hmc: 2 gradient calls per leapfrog + "
                             "1 value call per draw; mala: 3 gradient + 1 value, factorisation inside every dmvnorm), dense "
                             "identity mat-vecs, allocation per call")
    if "mode_b" in out:
        out["mode_b"]["what"] = ("optimised CPU, bit-identical draws: gradient reuse, no identity mat-vecs, SIMD (axpy-form) mat-vec, "
                                 "no allocation in the loop")
    if cfg["algo"] == "nuts":
        out["units"] = "leapfrogs AS THE REFERENCE EXECUTES THEM x dims / s (compare with executed.reference_equivalent_value)"
        if "mode_b" in out:
            out["mode_b"]["what"] = ("the memoised evaluation of every doubling on the CPU (oracle/mcmc_oracle.c: orc_nuts_memo -- the algorithm of "
                                     "nuts_memo.hpp): bit-identical draws, ~40 % fewer leap_frog calls; rate in reference-equivalent units")
    out["sample"] = (f"{out['mode_a']['chains']} chains (mode A) x {n_tot} draws of this workload, d={d}, OpenMP over chains on "
                     f"{cores} cores; rates extrapolate linearly in the chain count (chains are independent)")
    return out


def same_kernel(label, profiled_name):
    """Is the kernel rocprofv3 profiled (`void mi::name<args...>(params)`) the one the engine named (mi_mcmc_last_kernel: `name<args>`, possibly
    without trailing defaulted template arguments)?  The label up to its closing bracket must be a prefix of the profiled instantiation."""
    base = label.split("(")[0].strip()
    base = base[:-1] if base.endswith(">") else base
    return ("mi::" + base) in profiled_name or profiled_name.startswith(base)


def committed_pmc(cfg_id):
    """(round, path) of the committed counter passes of a config: profiles/r<N>_c<cfg>_pmc.json (the full-size run) and
    profiles/r<N>_c<cfg>_<C>chains_pmc.json (one GPU's share of an N-GPU split: what a rank of `--gpus N` launches)."""
    import glob
    import re
    cands = []
    for p in glob.glob(os.path.join(ROOT, "profiles", f"r*_c{cfg_id}_*pmc.json")):
        m = re.match(r"r(\d+)_c\d+_(\d+chains_)?pmc\.json$", os.path.basename(p))
        if m:
            cands.append((int(m.group(1)), p))
    return cands


def profiled_traffic(cfg_id, key, kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r<N>_c<cfg>_pmc.json, newest round first) --
    only from a profile of THIS workload (workload_key) and of THE KERNEL THAT JUST RAN (mi_mcmc_last_kernel): a figure measured on another
    kernel is not this kernel's traffic.  Returns (bytes, source) or (None, why)."""
    cands = committed_pmc(cfg_id)
    why = "no committed profile of this config"
    for _, p in sorted(cands, reverse=True):
        rel = os.path.relpath(p, ROOT)
        try:
            j = json.load(open(p))
            if j.get("workload_key") != list(key):
                why = f"{rel}: another workload shape {j.get('workload_key')}"
                continue
            prof_kernel = j["derived"]["kernel"]
            if not same_kernel(kernel_name, prof_kernel):
                why = f"{rel} profiled {prof_kernel.split('(')[0].replace('void mi::', '')}, this run launched {kernel_name}: not quoted"
                continue
            return j["derived"]["hbm_bytes_per_launch"], rel
        except (OSError, ValueError, KeyError):
            continue
    return None, why


def profiled_kernel_ms(cfg_id, key, kernel_name):
    """Average duration of the dominant kernel in the newest committed rocprofv3 --kernel-trace --stats pass of this workload and this kernel
    (profiles/r<N>_c<cfg>_pmc.json: derived.kernel_ms_avg, every launch of that run incl. its warm-up), or None."""
    for _, p in sorted(committed_pmc(cfg_id), reverse=True):
        try:
            j = json.load(open(p))
            if j.get("workload_key") == list(key) and same_kernel(kernel_name, j["derived"]["kernel"]):
                return float(j["derived"]["kernel_ms_avg"]), os.path.relpath(p, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def profiled_pipe_budget(cfg_id, key, kernel_name):
    """The combined-pipe budget of the dominant kernel -- (4 x VALU wave-instructions + MFMA busy cycles) / SIMD cycles -- from the newest
    committed counter pass of this workload AND this kernel (tools/summarize_prof.py: derived.pipe_budget); None if there is none."""
    cands = committed_pmc(cfg_id)
    for _, p in sorted(cands, reverse=True):
        try:
            j = json.load(open(p))
            if j.get("workload_key") == list(key) and same_kernel(kernel_name, j["derived"]["kernel"]) and "pipe_budget" in j["derived"]:
                return dict(j["derived"]["pipe_budget"], source=os.path.relpath(p, ROOT))
        except (OSError, ValueError, KeyError):
            continue
    return None


def measured_traffic(cfg_id, kernel, chains_arg):
    """HBM bytes per launch of `kernel`, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE: they do not fit one pass,
    /opt/skills/guides/MI355X_MICROARCH.md) of one step of this workload in a child process.  Units and gfx950 corrections as the
    guide prescribes: both counters are KiB; FETCH_SIZE is doubled for kernels whose reads are 16-byte-per-lane streams (the NUTS
    record rows), taken as reported otherwise; WRITE_SIZE as reported.  Returns (bytes, detail) or (None, reason)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="mi_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    base = kernel.split("<")[0]
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--config", str(cfg_id), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                   "--traffic", "none", "--no-ess"] + (["--chains", str(chains_arg)] if chains_arg else [])
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
            tot, disp = 0.0, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f, newline="")):
                    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                    if base in name and r.get("Counter_Name") == counter:
                        tot += float(r["Counter_Value"]); disp.add(r.get("Dispatch_Id"))
            if not disp:
                return None, f"{counter}: kernel {base} not in the counter output"
            per[counter] = tot / len(disp) * 1024.0
        factor = 2.0 if base.startswith("nuts_gauss_") else 1.0       # (every nuts_gauss_* kernel reads its records 16 bytes per lane)
        return per["FETCH_SIZE"] * factor + per["WRITE_SIZE"], {"read_bytes": per["FETCH_SIZE"] * factor, "write_bytes": per["WRITE_SIZE"],
                                                                   "fetch_factor": factor, "source": "rocprofv3 --pmc, this run"}
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
        return None, f"rocprofv3 pass failed: {type(e).__name__}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Ctx:
    pass


def extra_nuts_on_logistic(ctx):
    """Not a BASELINE config: mcmc::nuts on configs[2]'s OWN target (d = 512 logistic regression, N = 1024), the combination round 3 could
    only serve on the literal kernel.  One run of 32 768 chains x (4 burn-in + 4 kept) draws with dual averaging on the
    tiled kernel of mcmc_amd/csrc/nuts_lds.hpp, timed with events on the launch stream; a leapfrog is one fused evaluation (4 N d flop)."""
    import torch
    import mcmc_amd
    from mcmc_amd import synth
    d, n_rows, C, burn, keep = 512, 1024, 32768, 4, 4
    X, y = synth.logistic_problem(d, n_rows)
    dev = ctx.dev
    theta0 = torch.from_numpy(np.ascontiguousarray((synth.initial_states(C, d, seed=3) * 0.1).T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
    n_leap = torch.zeros(C, dtype=torch.int64, device=dev)
    n_exec = torch.zeros(C, dtype=torch.int64, device=dev)
    target = mcmc_amd.make_target(mcmc_amd.TARGET_LOGISTIC, d, mem=mcmc_amd.MEM_DEVICE, X=torch.from_numpy(X).to(dev), y=torch.from_numpy(y).to(dev))
    settings = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=burn, max_tree_depth=10, step_size=0.03)
    chains = mcmc_amd.make_chains(theta, C, draws=draws, n_leapfrogs=n_leap, n_leapfrogs_executed=n_exec, step_size=torch.zeros(C, dtype=torch.float64, device=dev),
                                  mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream
    ms = None
    for _ in range(2):                          # the first run loads the code object
        theta.copy_(theta0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        mcmc_amd.run("nuts", target, settings, chains, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
    # `value` and the roofline count what the device really computed (mi_chains.n_leapfrogs_executed: round 6, every doubling on a memoised trajectory --
    # one leapfrog per DISTINCT point); the leapfrogs the reference executes for the same draws are reported next to it
    leaps_ref = float(n_leap.double().sum().item())
    leaps = float(n_exec.double().sum().item())
    tflops = leaps * 4.0 * n_rows * d / (ms * 1e-3) / 1e12
    return {"workload": "mcmc::nuts on configs[2]'s target: d=512 Bayesian logistic regression (N=1024 synthetic rows), max_tree_depth=10, dual averaging, fp64",
            "chains": C, "draws": burn + keep, "ms": ms, "kernel": mcmc_amd.last_kernel(), "leapfrogs": leaps, "leapfrogs_as_the_reference_counts_them": leaps_ref,
            "value": leaps * d / (ms * 1e-3), "reference_equivalent_value": leaps_ref * d / (ms * 1e-3), "unit": "chain*dim*leapfrog-steps/s",
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
                         "flop_per_unit": 4 * n_rows},
            "note": "whole run incl. every chain's first evaluation and step-size search and the literal replay launch; chains are handed to the "
                    "8 192 chain slots of the persistent grid dynamically, their runs cut into pieces, see DESIGN.md section 4.14; value = executed leapfrogs (one per distinct point of a "
                    "doubling's trajectory), reference_equivalent_value = the leapfrogs mcmc::nuts executes for these draws / the same seconds"}


def extra_hmc_dense_precond(ctx):
    """Not a BASELINE config: mcmc::hmc with a DENSE precond_mat beyond d = 128 (ref: src/hmc.cpp:57-59,158-160,171,184), which ran on the
    literal kernel until round 5.  A dense Gaussian with d = 256, 65 536 chains, 16 leapfrog steps x (10 + 10) draws on
    logit_lds_kernel<.., DENSEM> (P, INV(M) and CHOL_LOWER(M) streamed through LDS); the shape of tools/dense_m_time.py.
    Events on the launch stream around the whole call, so the host's factorisation of M is inside `ms` (it is not inside the kernel)."""
    import torch
    import mcmc_amd
    from mcmc_amd import synth
    d, C, L, burn, keep = 256, 65536, 16, 10, 10
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    M = A @ A.T + np.diag(rng.uniform(0.4, 2.5, d))
    dev = ctx.dev
    theta0 = torch.from_numpy(np.ascontiguousarray((synth.initial_states(C, d, seed=3) * 0.3).T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
    target = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev), mem=mcmc_amd.MEM_DEVICE)
    settings = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.03, precond_mat=M)
    chains = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream
    ms = None
    for _ in range(2):                          # the first run loads the code object
        theta.copy_(theta0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        mcmc_amd.run("hmc", target, settings, chains, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
    n_draws = burn + keep
    flop = float(C) * n_draws * (L * 2 * (2.0 * d * d) + 3 * (2.0 * d * d))       # per leapfrog P x and Minv p; per draw L z and two kinetic energies
    tflops = flop / (ms * 1e-3) / 1e12
    return {"workload": "mcmc::hmc on a dense Gaussian, d=256, DENSE precond_mat, 16 leapfrog steps, fp64", "chains": C, "draws": n_draws, "ms": ms,
            "kernel": mcmc_amd.last_kernel(), "value": float(C) * d * L * n_draws / (ms * 1e-3), "unit": "chain*dim*leapfrog-steps/s",
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS,
                         "flop_per_unit": (L * 4.0 * d * d + 6.0 * d * d) / (d * L)},
            "note": "whole call incl. INV / CHOL_LOWER of M (round 6: on the device, mcmc_amd/csrc/linalg_device.hip; memoised after the first call) and the "
                    "literal replay launch"}


def extra_hmc_dense_d1024(ctx):
    """Not a BASELINE config: mcmc::hmc on a dense Gaussian BEYOND d = 512 (ref: src/hmc.cpp:40 -- n_vals is unrestricted; VERDICT r5 missing 5), which ran on
    the literal kernel (~1 % of the matrix peak) until round 6.  d = 1024, 65 536 chains, 16 leapfrog steps x (2 + 2) draws on gemm_step_kernel
    (mcmc_amd/csrc/gemm_samplers.hip, DESIGN.md section 4.18: the state in HBM, one fp64 matrix product W = P Theta per leapfrog step for all chains, kicks and
    drift in its epilogue).  Events on the launch stream around the whole call: the pack of P, the first evaluation, the normals and the accept step are inside `ms`."""
    import torch
    import mcmc_amd
    from mcmc_amd import synth
    d, C, L, burn, keep = 1024, 65536, 16, 2, 2
    dev = ctx.dev
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
    target = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev), mem=mcmc_amd.MEM_DEVICE)
    settings = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=0.02)
    chains = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream
    ms = None
    for _ in range(2):                          # the first run loads the code object
        theta.copy_(theta0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        mcmc_amd.run("hmc", target, settings, chains, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
    n_draws = burn + keep
    flop = float(C) * (n_draws * L + 1) * 2.0 * d * d
    tflops = flop / (ms * 1e-3) / 1e12
    return {"workload": "mcmc::hmc on a dense Gaussian, d=1024, identity precond_mat, 16 leapfrog steps, fp64", "chains": C, "draws": n_draws, "ms": ms,
            "kernel": mcmc_amd.last_kernel(), "value": float(C) * d * L * n_draws / (ms * 1e-3), "unit": "chain*dim*leapfrog-steps/s",
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS, "flop_per_unit": 2.0 * d},
            "note": "whole call; the kernel's own rocprofv3 average and counters: profiles/r6_gemm_kernel_stats.csv, profiles/r6_gemm_pmc.json"}


# ESS/sec legs: BASELINE's frozen settings of configs[2] (step_size 0.02: accept 0.99996, the chains barely move) and
# configs[4] (8 kept draws at M = I) give draws/sec but no ESS that is an estimate.  These legs run the SAME target and sampler, outside the
# timed region, at settings under which the chains converge (R-hat < 1.1); the values are frozen in BASELINE.md section 9.
CONVERGED = {
    3: dict(chains=65536, step_size=0.46, n_burnin_draws=100, n_keep_draws=400,
            what="mcmc::mala on configs[2]'s target, step_size 0.46 (accept ~ 0.58, ref: src/mala.cpp:170-173 -- the accept rule -- tuned on the "
                 "CPU oracle), 100 burn-in + 400 kept draws, 65 536 chains (400 kept draws of 262 144 chains do not fit one GPU's HBM)"),
    5: dict(chains=131072, step_size=0.12, n_burnin_draws=20, n_keep_draws=50, n_windows=3,
            what="mi_mcmc_hmc_run_mass_adapted on configs[4]'s shard: pooled diagonal mass, 3 windows, step_size 0.12, 50 kept draws "
                 "(NOT a reference mode: the reference has no mass adaptation; M = I at cond 1e4 does not mix in any affordable run)"),
}


def converged_leg(cfg_id, ctx):
    """One run of CONVERGED[cfg_id] on rank 0's GPU, timed host-side around the call (synchronize on both sides); returns the dict that
    goes under `converged` or None."""
    import torch
    import mcmc_amd
    if cfg_id not in CONVERGED:
        return None
    cfg, cv = dict(WORKLOADS[cfg_id]), CONVERGED[cfg_id]
    d, C, dev = cfg["d"], cv["chains"], ctx.dev
    n_keep = cv["n_keep_draws"]
    cfg_in = dict(cfg, chains=C) if cfg_id != 3 else cfg           # (the mala starts are spread by the FULL run's chain ids)
    kw, init = build_inputs(cfg_in, C, 0)
    kw_dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in kw.items()}
    theta = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    del init
    draws = torch.empty((n_keep, d, C), dtype=torch.float64, device=dev)
    n_accept = torch.zeros(C, dtype=torch.int64, device=dev)
    kind = mcmc_amd.TARGET_LOGISTIC if cfg["algo"] == "mala" else mcmc_amd.TARGET_GAUSS_DIAG
    target = mcmc_amd.make_target(kind, d, mem=mcmc_amd.MEM_DEVICE, **kw_dev)
    skw = dict(rng_seed_value=cfg["seed"], n_burnin_draws=cv["n_burnin_draws"], n_keep_draws=n_keep, step_size=cv["step_size"])
    if "n_leap_steps" in cfg:
        skw["n_leap_steps"] = cfg["n_leap_steps"]
    settings = mcmc_amd.default_settings(**skw)
    chains = mcmc_amd.make_chains(theta, C, draws=draws, n_accept=n_accept, mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if cfg_id == 5:
        mcmc_amd.hmc_mass_adapted(target, settings, chains, n_windows=cv["n_windows"], stream=stream)
    else:
        mcmc_amd.run(cfg["algo"], target, settings, chains, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = mcmc_amd.draw_stats(draws, n_keep, d, C, mem=mcmc_amd.MEM_DEVICE, stream=stream, want_acov=False)
    rhat = float(stats["rhat"].max())
    ess_chain = float(stats["ess"].min())
    out = {"what": cv["what"], "chains": C, "step_size": cv["step_size"], "n_burnin_draws": cv["n_burnin_draws"], "n_keep_draws": n_keep,
           "ms": dt * 1e3, "accept_rate": float(n_accept.double().mean().item()) / n_keep, "rhat_max": rhat,
           "ess_per_chain_min_over_dims": ess_chain, "ess_per_sec": ess_chain * C / dt,
           "ess_is_estimate": bool(n_keep >= 20 and rhat == rhat and rhat < 1.1), "kernel": mcmc_amd.last_kernel()}
    del draws, theta, n_accept, kw_dev, target, chains
    mcmc_amd.release_workspace()
    torch.cuda.empty_cache()
    return out


def time_collation(args, ctx, cfg, algo, target, skw, theta, theta0, draws, chain0, C, total, n_keep, d, stream, barrier, eps_out):
    """Times the collation of draws_out through the C ABI on every rank (barrier + synchronize on both sides, so the figure is the slowest
    rank's): blocking (mi_mcmc_allgather_draws_rank_major) and overlapped with the sampling (4 chunks, _begin / _wait).  Returns the dict that
    goes under `collation` (identical on every rank).  Under BENCH_TEST_SHARE_GPU (the ranks share ONE device, which RCCL refuses inside one
    communicator) every rank collates its own shard over a ONE-rank communicator: the same C calls, no bytes between devices -- labelled."""
    import torch
    import mcmc_amd
    from mcmc_amd import dist as mdist
    world, rank, dev, share = ctx.world, ctx.rank, ctx.dev, ctx.share
    solo = share or world == 1
    comm = mdist.RcclComm(solo=solo)
    n_total = C if solo else total                           # chains in the receive buffer
    out = {"abi": "mi_mcmc_allgather_draws_rank_major; mi_mcmc_allgather_draws_begin / _wait (include/mi_mcmc.h)",
           "rccl_ranks": comm.world, "layout": "rank-major [G][n_keep][d][C/G] (SURVEY 8(e)), one ncclAllGather, no staging buffer",
           "bytes_received_per_rank": n_keep * d * n_total * 8, "bytes_sent_per_rank": n_keep * d * C * 8}
    if solo and world > 1:
        out["note"] = ("BENCH_TEST_SHARE_GPU: the ranks share one device, so each collates its OWN shard over a one-rank RCCL communicator "
                       "(the C-ABI calls run, nothing crosses xGMI): NOT a multi-GPU collation time")
    recv = torch.empty(n_keep * d * max(n_total, 1), dtype=torch.float64, device=dev)
    loc = draws[:, :, :C] if C == draws.shape[2] else draws[:, :, :C].contiguous()
    mdist.collate_rank_major(comm, loc, n_total, out=recv, stream=stream)      # warm-up: the communicator's first collective sets its rings up
    reps = []
    for _ in range(3):
        barrier()
        tc = time.perf_counter()
        mdist.collate_rank_major(comm, loc, n_total, out=recv, stream=stream)
        barrier()
        reps.append((time.perf_counter() - tc) * 1e3)
    out["blocking_ms"] = float(np.median(reps))
    out["blocking_ms_reps"] = reps
    out["GBps"] = out["bytes_received_per_rank"] / (out["blocking_ms"] * 1e-3) / 1e9
    out["busbw_GBps"] = out["GBps"] * (comm.world - 1) / comm.world          # the all-gather convention: what crossed the links, per rank
    # the rank's own shard sits where the layout says (equal shards: block `rank` of the receive buffer)
    if C > 0 and (solo or total % world == 0):
        r_ = 0 if solo else rank
        blk = recv.view(-1)[r_ * n_keep * d * C:(r_ + 1) * n_keep * d * C].view(n_keep, d, C)
        out["own_shard_in_place"] = bool(torch.equal(blk, loc))
    # ... and OVERLAPPED with the sampling: the same run in four chunks chained through mi_chains.draw0 (bit-identical to one call:
    # tests/test_gpu_resume.py); a nuts continuation starts after its adaptation window and takes the adapted step sizes back in
    K = 4
    ok = n_keep >= K and (algo in ("hmc", "mala") or (algo == "nuts" and cfg.get("n_adapt_draws", 0) <= cfg["n_burnin_draws"]))
    if ok:
        bounds = [(n_keep * i) // K for i in range(K + 1)]
        slabs = [torch.empty((bounds[i + 1] - bounds[i], d, max(C, 1)), dtype=torch.float64, device=dev) for i in range(K)]
        recv.zero_()
        theta.copy_(theta0)
        barrier()
        tc = time.perf_counter()
        handles = []
        for i in range(K):
            s_i = mcmc_amd.default_settings(**dict(skw, n_burnin_draws=cfg["n_burnin_draws"] if i == 0 else 0, n_keep_draws=bounds[i + 1] - bounds[i]))
            ch_i = mcmc_amd.make_chains(theta, C, chain0=chain0, draws=slabs[i], mem=mcmc_amd.MEM_DEVICE, step_size=eps_out if algo == "nuts" else None,
                                        draw0=0 if i == 0 else cfg["n_burnin_draws"] + bounds[i])
            if C > 0:
                mcmc_amd.run(algo, target, s_i, ch_i, stream=stream)
            handles.append(mdist.collate_begin(comm, slabs[i], n_total, bounds[i], n_keep, recv, producer_stream=stream))
        for i, h in enumerate(handles):
            mdist.collate_wait(h, consumer_stream=stream, block_host=(i == K - 1))
        barrier()
        out["overlapped_total_ms"] = (time.perf_counter() - tc) * 1e3
        out["overlapped_chunks"] = K
        if C > 0 and (solo or total % world == 0):
            r_ = 0 if solo else rank
            blk = recv.view(-1)[r_ * n_keep * d * C:(r_ + 1) * n_keep * d * C].view(n_keep, d, C)
            out["overlapped_equals_blocking_run"] = bool(torch.equal(blk, loc))      # the chunked run reproduces the one-call run's rows
        del slabs
    else:
        out["overlapped_total_ms"] = None
        out["overlapped_why_not"] = "needs n_keep >= 4 and, for nuts, an adaptation window inside the burn-in"
    del recv
    comm.close()
    return out


def measure(cfg_id, steps, warmup, args, ctx, headline, chains_override=None, want_traffic=True):
    """Times `steps` steps of one BASELINE config on this rank's GPU (barrier + synchronize on both sides, max over ranks).
    Returns the result dict on rank 0 (None elsewhere)."""
    import torch
    import mcmc_amd
    from mcmc_amd import dist as mdist
    world, rank, dev, dist, share = ctx.world, ctx.rank, ctx.dev, ctx.dist, ctx.share
    cfg = dict(WORKLOADS[cfg_id])
    d, algo = cfg["d"], cfg["algo"]
    scaling = args.scaling or "strong"     # north_star: the chain total is fixed as N grows (at N = 1 the two coincide)
    n_chains_arg = chains_override or args.chains
    if scaling == "weak":
        C = n_chains_arg or cfg["chains"]
        chain0, total = rank * C, (n_chains_arg or cfg["chains"]) * world
    else:
        total = n_chains_arg or cfg["chains"]
        chain0, C = mdist.shard_bounds(total, world, rank)              # balanced, no chain dropped
    n_keep, n_tot = cfg["n_keep_draws"], cfg["n_burnin_draws"] + cfg["n_keep_draws"]

    kw, init = build_inputs(cfg, max(C, 1), chain0)
    kw_dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in kw.items()}
    theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((n_keep, d, max(C, 1)), dtype=torch.float64, device=dev)
    n_accept = torch.zeros(max(C, 1), dtype=torch.int64, device=dev)
    n_leap = torch.zeros(max(C, 1), dtype=torch.int64, device=dev)
    n_exec = torch.zeros(max(C, 1), dtype=torch.int64, device=dev)
    eps_out = torch.zeros(max(C, 1), dtype=torch.float64, device=dev)

    kind = {"hmc": mcmc_amd.TARGET_GAUSS_DENSE if d <= 128 else mcmc_amd.TARGET_GAUSS_DIAG,
            "nuts": mcmc_amd.TARGET_GAUSS_DENSE, "mala": mcmc_amd.TARGET_LOGISTIC}[algo]
    target = mcmc_amd.make_target(kind, d, mem=mcmc_amd.MEM_DEVICE, **kw_dev)
    skw = dict(rng_seed_value=cfg["seed"], n_burnin_draws=cfg["n_burnin_draws"], n_keep_draws=n_keep)
    for k in ("n_leap_steps", "step_size", "n_adapt_draws", "max_tree_depth"):
        if k in cfg:
            skw[k] = cfg[k]
    settings = mcmc_amd.default_settings(**skw)
    chains = mcmc_amd.make_chains(theta, C, chain0=chain0, draws=draws, n_accept=n_accept, n_leapfrogs=n_leap,
                                  n_leapfrogs_executed=n_exec, step_size=eps_out, mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream

    def one_step():
        theta.copy_(theta0)                     # same start every step (device-to-device, untimed by events)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()                            # torch's current stream IS the launch stream (passed to the engine below)
        if C > 0:
            mcmc_amd.run(algo, target, settings, chains, stream=stream)
        ev1.record()
        return ev0, ev1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    events = [one_step() for _ in range(steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    kernel_name = mcmc_amd.last_kernel() if C > 0 else cfg["kernel"]   # what the engine actually launched (mi_mcmc_last_kernel)

    # units of this rank per step: EXECUTED leapfrog steps x dims (hmc, nuts) or draws x dims (mala).  nuts on the memoised kernel
    # (nuts_memo.hpp) computes every distinct state of a doubling once: it executes fewer leapfrogs than mcmc::nuts does for the same draws --
    # `value` counts only what the device really computed (mi_chains.n_leapfrogs_executed); the reference's own count is reported next to it
    units_ref_rank = None
    if algo == "mala":
        units_rank = float(C) * d * n_tot
    else:
        units_rank = float(n_exec[:C].double().sum().item()) * d if C else 0.0
        units_ref_rank = float(n_leap[:C].double().sum().item()) * d if C else 0.0
        if algo == "hmc" and C:
            assert int(n_leap[0].item()) == n_tot * cfg["n_leap_steps"] == int(n_exec[0].item())
    units_all = units_rank
    if dist is not None:
        t = torch.tensor([elapsed, units_rank], dtype=torch.float64, device=dev if not share else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, units_all = float(tmax[0].item()), float(t[1].item())

    # The ONE exchange the path has (north_star: "a single RCCL all-gather over xGMI to collate draws_out"; what is collated is the rows of
    # ref: src/hmc.cpp:196-204), ALWAYS timed when there is more than one rank ("report both with and without") and THROUGH THE
    # C ABI -- the product's own collation (include/mi_mcmc.h), not torch's: mi_mcmc_allgather_draws_rank_major (blocking: sampling, then one
    # gather) and mi_mcmc_allgather_draws_begin / _wait (the run cut into 4 chunks through mi_chains.draw0, chunk k's slab travelling on the
    # library's communication stream while chunk k + 1 samples).  Outside `value`'s timed region; printed next to it.
    collation = None
    if headline and dist is not None and (world > 1 or args.collate):
        collation = time_collation(args, ctx, cfg, algo, target, skw, theta, theta0, draws, chain0, C, total if scaling == "strong" else C * world,
                                   n_keep, d, stream, barrier, eps_out)

    # ESS/sec (second half of BASELINE.json's metric): Geyer initial-positive-sequence ESS, min over dims, autocovariances pooled
    # over ALL chains of this rank by the device reducer (mi_mcmc_draw_stats, no D2H of the draws); outside the timed region, its own
    # time reported next to it
    ess_total_rank, rhat_max, reducer_ms = 0.0, float("nan"), None
    if C > 0 and rank == 0 and not args.no_ess and chains_override is None:
        if not getattr(ctx, "stats_warm", False):           # untimed warm-up of the reducer's kernels (code-object load) on a tiny slab
            mcmc_amd.draw_stats(torch.randn((8, 2, 64), dtype=torch.float64, device=dev), 8, 2, 64, mem=mcmc_amd.MEM_DEVICE,
                                stream=stream, want_acov=False)
            mcmc_amd.draw_stats(torch.randn((40, 2, 64), dtype=torch.float64, device=dev).cumsum(0), 40, 2, 64, mem=mcmc_amd.MEM_DEVICE,
                                stream=stream, want_acov=False)
            ctx.stats_warm = True
        torch.cuda.synchronize()
        tr = time.perf_counter()
        stats = mcmc_amd.draw_stats(draws, n_keep, d, C, mem=mcmc_amd.MEM_DEVICE, stream=stream, want_acov=False)
        torch.cuda.synchronize()
        reducer_ms = (time.perf_counter() - tr) * 1e3
        ess_total_rank = float(stats["ess"].min()) * C
        rhat_max = float(stats["rhat"].max())
    acc_rate = float(n_accept[:C].double().mean().item()) / n_keep if C else float("nan")
    value = units_all * steps / elapsed

    out = None
    if rank == 0:
        fpu = flop_per_unit(cfg)
        k_ms = float(np.mean(kernel_ms))
        achieved = units_rank * fpu / (k_ms * 1e-3) / 1e12
        key = (cfg_id, C, d, n_tot)
        traffic, traffic_src = None, None
        want_pmc = want_traffic and (args.traffic == "all" or (args.traffic == "headline" and headline))
        if want_pmc and world == 1:
            traffic, traffic_src = measured_traffic(cfg_id, kernel_name, n_chains_arg)
        if traffic is None and want_traffic:
            why = traffic_src
            traffic, traffic_src = profiled_traffic(cfg_id, key, kernel_name)
            if traffic is None and want_pmc:
                traffic_src = f"{why}; {traffic_src}"
        out = {
            "metric": cfg["metric"], "value": value, "unit": cfg["unit"],
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "chains_per_gpu": C, "chains_total": total, "d": d,
                       "n_burnin_draws": cfg["n_burnin_draws"], "n_keep_draws": n_keep,
                       "parallelism": f"chains sharded x{world} by global chain id, no data-path collective",
                       "accept_rate": acc_rate},
            "roofline": {"bound": cfg["bound"], "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms": k_ms, "flop_per_unit": fpu},
        }
        # achieved = max(bytes term, flops term), reported with BOTH terms.  The bytes term is the 32 B / unit streaming
        # model (read theta, p + write theta, p per chain x dim x step) against the 8 TB/s HBM peak; a kernel that keeps the state
        # register-resident moves less than the model, so its bytes term can exceed 1 -- it is a model, the counter figure below is
        # what the launch moved.  `frac` stays the term of the bound SURVEY 8(d) assigns the config.
        units_per_s = units_rank / (k_ms * 1e-3)
        out["roofline"]["flops_term"] = {"achieved_TFLOPs": achieved, "peak_TFLOPs": FP64_PEAK_TFLOPS, "frac": achieved / FP64_PEAK_TFLOPS}
        bfrac = units_per_s * 32 / 8e12
        out["roofline"]["bytes_model_term"] = {"bytes_per_unit": 32, "achieved_TBps": units_per_s * 32 / 1e12, "peak_TBps": 8.0,
                                               "frac": bfrac if bfrac <= 1.0 else None}
        if bfrac > 1.0:
            out["roofline"]["bytes_model_term"]["why_null"] = ("n/a: the state is register-resident, nothing streams per unit -- the model's "
                                                               "32 B / unit are not moved (see hbm_TBps for what the launch moved)")
        out["roofline"]["kernel_ms_source"] = "mean of the HIP-event durations of the timed steps on the launch stream (warm-up excluded)"
        pk_ms, pk_src = profiled_kernel_ms(cfg_id, key, kernel_name) if want_traffic else (None, None)
        if pk_ms:        # the same fraction from the committed rocprofv3 average (which includes that run's warm-up launch): the figure to quote next to profiles/
            out["roofline"]["frac_rocprof"] = {"kernel_ms_avg": pk_ms, "frac": units_rank * fpu / (pk_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "source": pk_src}
        pb = profiled_pipe_budget(cfg_id, key, kernel_name) if want_traffic else None
        if pb is not None:           # what the kernel's instruction mix allows (fp64 VALU and MFMA share a pipe): frac / pipe_frac is the slack
            out["roofline"]["pipe_budget"] = pb
        if traffic is not None:      # HBM side of the same launch
            out["roofline"]["hbm_TBps"] = traffic / (k_ms * 1e-3) / 1e12
            out["roofline"]["hbm_frac_of_8TBps"] = out["roofline"]["hbm_TBps"] / 8.0
        for k in ("n_leap_steps", "step_size", "n_adapt_draws", "max_tree_depth", "n_rows"):
            if k in cfg:
                out["config"][k] = cfg[k]
        if algo != "mala" and C:
            out["config"]["leapfrogs_per_chain_mean"] = units_rank / d / C
        if algo == "nuts" and C and world == 1:
            out["executed"] = {"leapfrogs_executed_per_chain_mean": units_rank / d / C,
                               "leapfrogs_reference_per_chain_mean": units_ref_rank / d / C,
                               "executed_over_reference": units_rank / units_ref_rank if units_ref_rank else None,
                               "reference_equivalent_value": units_ref_rank * steps / elapsed,
                               "note": "`value` and the roofline count EXECUTED leapfrogs only.  The memoised kernel (nuts_memo.hpp) produces the draws "
                                       "mcmc::nuts produces -- bit for bit -- while computing each distinct state of a doubling once; "
                                       "reference_equivalent_value = the leapfrogs mcmc::nuts executes for these draws x dims / the same seconds"}
        if algo == "nuts":
            out["config"]["adapted_step_size_mean"] = float(eps_out[:C].mean().item())
        if reducer_ms is not None:
            step_s = elapsed / steps
            out["ess_per_sec"] = ess_total_rank * world / step_s
            out["ess_per_sec_incl_reducer"] = ess_total_rank * world / (step_s + reducer_ms * 1e-3)
            out["ess_reducer_ms"] = reducer_ms
            out["ess_note"] = (f"min-over-dims Geyer-IPS ESS of the {n_keep} kept draws (autocovariance pooled over all chains of rank 0 on "
                               "the device), x chains x ranks, / seconds per step; _incl_reducer adds the time of mi_mcmc_draw_stats itself")
            out["rhat_max"] = rhat_max
            # a number is not an estimate: few kept draws or chains that have not converged (R-hat far from 1) make ESS meaningless
            out["ess_is_estimate"] = bool(n_keep >= 20 and rhat_max == rhat_max and rhat_max < 1.1)
            if not out["ess_is_estimate"]:
                out["ess_caveat"] = (f"NOT an estimate: n_keep = {n_keep}" + (" < 20" if n_keep < 20 else "") +
                                     f", rhat_max = {rhat_max:.3g}" + (" (chains not converged)" if not rhat_max < 1.1 else ""))
        if headline and cfg_id == 5 and C > 1 and not args.no_ess:
            # the second half of BASELINE.json's metric on this target is decided by the mass matrix, not by the kernel:
            # the same workload through mi_mcmc_hmc_run_mass_adapted (NOT a reference mode), outside the timed region
            theta.copy_(theta0)
            s_ad = mcmc_amd.default_settings(**dict(skw, step_size=0.12))
            torch.cuda.synchronize()
            ta = time.perf_counter()
            mass = mcmc_amd.hmc_mass_adapted(target, s_ad, chains, n_windows=3, stream=stream)
            torch.cuda.synchronize()
            t_ad = time.perf_counter() - ta
            st_ad = mcmc_amd.draw_stats(draws, n_keep, d, C, mem=mcmc_amd.MEM_DEVICE, stream=stream, want_acov=False)
            out["mass_adapted"] = {"what": "mi_mcmc_hmc_run_mass_adapted, pooled diagonal mass, 3 windows, step_size 0.12 (non-reference mode)",
                                   "ms": t_ad * 1e3, "ess_per_sec": float(st_ad["ess"].min()) * C * world / t_ad,
                                   "accept_rate": float(n_accept[:C].double().mean().item()) / n_keep,
                                   "mass_over_precision_range": [float((mass / kw["prec"]).min()), float((mass / kw["prec"]).max())]}
        if collation is not None:
            step_s = elapsed / steps
            units_step = units_all                          # units of ALL ranks in one step
            out["value_incl_collation"] = units_step / (step_s + collation["blocking_ms"] * 1e-3)
            out["value_overlapped"] = (units_step / (collation["overlapped_total_ms"] * 1e-3)) if collation.get("overlapped_total_ms") else None
            out["collate_GBps"] = collation["GBps"]
            out["collation"] = collation
    del draws, theta, theta0, n_accept, n_leap, n_exec, eps_out, kw_dev, target, chains
    mcmc_amd.release_workspace()
    torch.cuda.empty_cache()
    return out, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=sorted(WORKLOADS),
                    help="one BASELINE config only; default: the headline (2) with --steps/--warmup, then configs 3, 4, 5 at "
                         "--other-steps (one GPU only), attached as other_configs")
    ap.add_argument("--other-steps", type=int, default=2)
    ap.add_argument("--other-warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong when WORLD_SIZE > 1 (north_star), n/a at one GPU")
    ap.add_argument("--chains", type=int, default=None, help="chains: total (strong) / per GPU (weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra (non-BASELINE) measurements of the default run")
    ap.add_argument("--traffic", choices=["headline", "all", "none"], default="all",
                    help="measure roofline.traffic with two rocprofv3 --pmc child runs per config (one GPU only); headline: only the headline "
                         "config that way, the others from the committed profiles/ of the same kernel; none: committed profiles only")
    ap.add_argument("--proxy-scaling", action="store_true", help="with --config N: also time chains/2, /4, /8 on this GPU (scaling_proxy)")
    ap.add_argument("--no-proxy", action="store_true", help="skip scaling_proxy in the default run")
    ap.add_argument("--converged", action="store_true", help="with --config 3 | 5: also run that config's converged ESS leg (the default run always does)")
    ap.add_argument("--collate", action="store_true",
                    help="one GPU: run the C-ABI collation over a one-rank RCCL communicator too (with N > 1 ranks it always runs)")
    args = ap.parse_args()

    import torch

    # --gpus N without a launcher: this process IS the launcher (it used to run ONE rank and print n_gpus 1)
    share = os.environ.get("BENCH_TEST_SHARE_GPU") == "1"        # test only: ranks share the visible GPU(s) over gloo
    n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if n_vis < args.gpus and not share:
            print(f"bench.py: --gpus {args.gpus} but only {n_vis} GPU(s) visible: refusing to report an N={args.gpus} line", file=sys.stderr)
            raise SystemExit(3)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"# bench.py: spawning {args.gpus} ranks (one per GPU): {' '.join(cmd[2:9])} ...", file=sys.stderr)
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import mcmc_amd
    from mcmc_amd import dist as mdist

    ctx = Ctx()
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = int(os.environ.get("RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.gpus != ctx.world:
        if ctx.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ctx.world}: the line would not be the run that was asked for", file=sys.stderr)
        raise SystemExit(2)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not share and local_rank >= n_vis:
        print(f"bench.py: LOCAL_RANK {local_rank} but only {n_vis} GPU(s) visible", file=sys.stderr)
        raise SystemExit(3)
    # BENCH_TEST_SHARE_GPU=1 (test only): gloo backend and LOCAL_RANK folded onto the visible devices, to exercise the N>1 path on a 1-GPU box
    ctx.share = share
    ctx.dev = mdist.bind_device(None if ctx.share else local_rank)
    ctx.dist = None
    ranks_seen, backend = 1, None
    if ctx.world > 1 or args.collate:          # (--collate at one GPU: a 1-rank RCCL group, so that the collation code runs on the box that has one)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if ctx.share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=ctx.dev)
        ctx.dist = dist
        ranks_seen, backend = dist.get_world_size(), ("gloo (BENCH_TEST_SHARE_GPU)" if ctx.share else "nccl (RCCL)")
        # every rank on its own device (unless the test mode folds them): the N-GPU line must be N GPUs
        mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device="cpu" if ctx.share else ctx.dev)
        seen = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(seen, mine)
        n_distinct = len({int(t.item()) for t in seen})
        if not ctx.share and n_distinct != ctx.world:
            raise SystemExit(f"bench.py: {ctx.world} ranks on {n_distinct} distinct devices")

    head_id = args.config or 2
    out, cfg = measure(head_id, args.steps, args.warmup, args, ctx, headline=True)
    if ctx.rank == 0:
        out["ranks"] = {"world_size": ranks_seen, "backend": backend, "devices_visible": n_vis}
    single = args.config is None and ctx.world == 1 and args.chains is None
    if single:
        # the other single-GPU BASELINE configs under the same clock, at a reduced step count
        others = []
        for cid in (3, 4, 5):
            o, ocfg = measure(cid, args.other_steps, args.other_warmup, args, ctx, headline=False)
            ent = {"config_id": cid, "workload": o["config"]["workload"], "metric": o["metric"], "unit": o["unit"],
                   "value": o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                   "chains": o["config"]["chains_per_gpu"], "accept_rate": o["config"]["accept_rate"],
                   "roofline": o["roofline"],
                   **({k: o[k] for k in ("ess_per_sec", "ess_per_sec_incl_reducer", "ess_reducer_ms", "rhat_max", "ess_is_estimate",
                                         "ess_caveat", "executed") if k in o})}
            if not args.no_ess and cid in CONVERGED:
                # ESS/sec of this config = the converged leg's; the frozen-settings run keeps its (non-)estimate next to it
                ent["frozen_run_ess"] = {k: ent.pop(k) for k in ("ess_per_sec", "ess_per_sec_incl_reducer", "ess_reducer_ms", "rhat_max",
                                                                 "ess_is_estimate", "ess_caveat") if k in ent}
                leg = converged_leg(cid, ctx)
                ent["converged"] = leg
                ent["ess_per_sec"], ent["rhat_max"], ent["ess_is_estimate"] = leg["ess_per_sec"], leg["rhat_max"], leg["ess_is_estimate"]
                ent["ess_source"] = "converged leg (outside the timed region; `value` and ms_per_step are the frozen BASELINE settings)"
            if not args.no_cpu_baseline:
                ent["cpu_baseline"] = cpu_baseline(cid, ocfg)
                gv = ent.get("executed", {}).get("reference_equivalent_value") or ent["value"]      # nuts: same work on both sides
                ent["gpu_over_cpu"] = {k: gv / ent["cpu_baseline"][k]["value"] for k in ("mode_a", "mode_b") if k in ent["cpu_baseline"]}
            others.append(ent)
        out["other_configs"] = others
        if not args.no_extra:
            out["extra"] = {"nuts_on_configs2_target": extra_nuts_on_logistic(ctx)}
            try:                                # (added after the round's last GPU minute: a failure here must not cost the line)
                out["extra"]["hmc_dense_precond_d256"] = extra_hmc_dense_precond(ctx)
            except Exception as e:              # noqa: BLE001
                out["extra"]["hmc_dense_precond_d256"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["extra"]["hmc_dense_d1024"] = extra_hmc_dense_d1024(ctx)
            except Exception as e:              # noqa: BLE001
                out["extra"]["hmc_dense_d1024"] = {"error": f"{type(e).__name__}: {e}"}
    elif ctx.rank == 0 and ctx.world == 1 and args.chains is None and head_id in CONVERGED and args.converged:
        out["converged"] = converged_leg(head_id, ctx)
    if (single and not args.no_proxy) or (args.proxy_scaling and ctx.world == 1):
        # ONE-GPU PROXY of the strong-scaling curve: the share of one GPU at N = 1, 2, 4, 8 -- total / N chains -- timed here
        # under the same clock.  It prices the kernel's few-chain launch shapes only: no collation, no launch skew between ranks, no xGMI.
        proxy = {"note": "ONE GPU: ms of one rank's share (chains_total / N) of the run -- what N GPUs would each spend in the sampling "
                         "kernel; a proxy, not a multi-GPU measurement (no collation, no rank skew)", "configs": []}
        for cid in ([2, 4] if single else [head_id]):
            total = WORKLOADS[cid]["chains"]
            rows = []
            for n in (1, 2, 4, 8):
                if n == 1 and cid == head_id:
                    ms, kern = out["ms_per_step"], out["roofline"]["kernel"]
                elif n == 1 and single:
                    ent = [e for e in out["other_configs"] if e["config_id"] == cid][0]
                    ms, kern = ent["ms_per_step"], ent["roofline"]["kernel"]
                else:
                    o, _ = measure(cid, 3, 1, args, ctx, headline=False, chains_override=total // n, want_traffic=False)
                    ms, kern = o["ms_per_step"], o["roofline"]["kernel"]
                rows.append({"n_gpus_modelled": n, "chains_per_gpu": total // n, "ms_per_step": ms, "kernel": kern})
            for r in rows:
                r["speedup_vs_1"] = rows[0]["ms_per_step"] / r["ms_per_step"]
            proxy["configs"].append({"config_id": cid, "chains_total": total, "curve": rows})
        if ctx.rank == 0:
            out["scaling_proxy"] = proxy
    if ctx.rank == 0:
        if not args.no_cpu_baseline:             # rank 0's host cores, at any N
            out["cpu_baseline"] = cpu_baseline(head_id, cfg)
            gv = out.get("executed", {}).get("reference_equivalent_value") or out["value"]
            out["gpu_over_cpu"] = {k: gv / out["cpu_baseline"][k]["value"] for k in ("mode_a", "mode_b") if k in out["cpu_baseline"]}
        print(json.dumps(out))
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
