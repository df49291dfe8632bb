#!/usr/bin/env python3
"""bench.py -- leapfrog-steps/sec (chains x dims x steps / s) of the many-chain HMC hot path.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): mcmc::hmc on a d=128 correlated
Gaussian (P = A A^T / d + I, analytic gradient), 65 536 chains per GPU, fp64, step_size 0.05,
n_leap_steps 16, 100 burn-in + 100 kept draws.  One "step" = one mi_mcmc_hmc_run call = that whole
sampling run for every chain of the rank, with target, initial states and output buffers already
resident in HBM.  Chains shard across ranks by global chain id with no data-path collective
(scaling "weak": 65 536 chains per GPU; pass --scaling strong to keep 65 536 chains in total).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- fp64 matrix-core bound of the fused HMC kernel: algorithmic flops per launch
                  (264 flop per chain.dim.leapfrog at d=128) / HIP-event duration
  cpu_baseline -- the CPU oracle (oracle/liboracle.so, a port of the reference algorithm) timed on
                  this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_MATRIX_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix (= vector) peak; not in the in-image guide
# HBM bytes per launch of the default workload, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
# command (profiles/r1_hmc_pmc.json): (2.3 + 34.0) GB. Not re-measured here (counters cannot be read in-process);
# reported only for the exact profiled configuration, null otherwise.  FETCH_SIZE is uncorrected (8-B-per-lane loads).
PROFILED_TRAFFIC_BYTES = {(65536, 128, 16, 100, 100): 3.64e10}

WORKLOAD = dict(d=128, chains_per_gpu=65536, n_leap_steps=16, step_size=0.05,
                n_burnin_draws=100, n_keep_draws=100, seed=2024)


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


def cpu_baseline(cfg, prec):
    """Oracle (port of src/hmc.cpp, reference-faithful work profile) on the host cores."""
    import orc   # tests/orc.py: ctypes binding of oracle/liboracle.so
    from mcmc_amd import synth
    cores = usable_cores()
    n_chains = 128 * cores      # ~10 s of CPU work on the box's 16 granted cores (0.09 s per chain of this workload)
    d = cfg["d"]
    init = synth.initial_states(n_chains, d, seed=3)
    tgt = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    st = orc.make_settings(seed=cfg["seed"], n_burnin=cfg["n_burnin_draws"], n_keep=cfg["n_keep_draws"],
                           n_leap=cfg["n_leap_steps"], step=cfg["step_size"], W=4)
    t0 = time.perf_counter()
    _, info = orc.run_many(orc.ALGO_HMC, tgt, init, st, n_threads=cores, want_draws=False)
    dt = time.perf_counter() - t0
    units = float(info["n_leap"].sum()) * d
    return {"value": units / dt, "unit": "chain*dim*leapfrog-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_chains} chains x {cfg['n_burnin_draws'] + cfg['n_keep_draws']} draws x "
                      f"{cfg['n_leap_steps']} leapfrogs, d={d}, {dt:.2f}s wall, OpenMP over chains"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (weak) / total (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collate", action="store_true",
                    help="also time the RCCL all-gather of the last kept draw (not part of `value`)")
    args = ap.parse_args()

    import torch
    import mcmc_amd
    from mcmc_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # BENCH_TEST_SHARE_GPU=1 (test only): all ranks on GPU 0 with the gloo backend, to exercise the N>1 code path on a 1-GPU box
    share = os.environ.get("BENCH_TEST_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    cfg = dict(WORKLOAD)
    d = cfg["d"]
    if args.scaling == "weak":
        C = args.chains or cfg["chains_per_gpu"]
        chain0 = rank * C
    else:
        total = args.chains or cfg["chains_per_gpu"]
        C = total // world
        chain0 = rank * C
    n_keep, n_tot = cfg["n_keep_draws"], cfg["n_burnin_draws"] + cfg["n_keep_draws"]

    dev = torch.device("cuda", local_rank)
    prec_h = synth.dense_gaussian_precision(d)
    prec = torch.from_numpy(prec_h).to(dev)
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3, chain0=chain0).T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((n_keep, d, C), dtype=torch.float64, device=dev)
    n_accept = torch.zeros(C, dtype=torch.int64, device=dev)
    n_leap = torch.zeros(C, dtype=torch.int64, device=dev)

    target = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE)
    settings = mcmc_amd.default_settings(rng_seed_value=cfg["seed"], n_burnin_draws=cfg["n_burnin_draws"],
                                         n_keep_draws=n_keep, n_leap_steps=cfg["n_leap_steps"],
                                         step_size=cfg["step_size"])
    chains = mcmc_amd.make_chains(theta, C, chain0=chain0, draws=draws, n_accept=n_accept,
                                  n_leapfrogs=n_leap, mem=mcmc_amd.MEM_DEVICE)
    stream = torch.cuda.current_stream().cuda_stream

    def one_step():
        theta.copy_(theta0)                     # same start every step (device-to-device, untimed by events)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        mcmc_amd.run("hmc", target, settings, chains, stream=stream)
        ev1.record()
        return ev0, ev1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    events = [one_step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    collate_ms = None
    if args.collate and dist is not None:
        last = draws[-1].contiguous()
        gathered = torch.empty((world * last.shape[0],) + tuple(last.shape[1:]), dtype=last.dtype, device=dev)
        barrier()
        tc = time.perf_counter()
        dist.all_gather_into_tensor(gathered, last)
        barrier()
        collate_ms = (time.perf_counter() - tc) * 1e3

    # ESS/sec (second half of BASELINE.json's metric): Geyer initial-positive-sequence ESS, min over dims, autocovariances pooled
    # over ALL chains of this rank by the device reducer (mi_mcmc_draw_stats, no D2H of the draws); outside the timed region
    stats = mcmc_amd.draw_stats(draws, n_keep, d, C, mem=mcmc_amd.MEM_DEVICE, stream=stream)
    ess_total_rank = float(stats["ess"].min()) * C
    rhat_max = float(stats["rhat"].max())

    leap_per_chain = int(n_leap[0].item())
    acc_rate = float(n_accept.double().mean().item()) / n_keep
    assert leap_per_chain == n_tot * cfg["n_leap_steps"]
    units_per_step_rank = float(C) * d * leap_per_chain
    units_per_step = units_per_step_rank * world
    value = units_per_step * args.steps / elapsed

    if rank == 0:
        flop_per_unit = 2 * d + 8                      # dense mat-vec with gradient reuse + leapfrog
        k_ms = float(np.mean(kernel_ms))
        achieved = units_per_step_rank * flop_per_unit / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "leapfrog-steps/sec (chains*dims*steps/s), HMC d=128 correlated Gaussian, 65536 chains/GPU",
            "value": value, "unit": "chain*dim*leapfrog-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: mcmc::hmc, d=128 dense-precision Gaussian "
                                   "(P=AA^T/d+I), analytic grad, fp64",
                       "chains_per_gpu": C, "chains_total": C * world, "d": d,
                       "n_leap_steps": cfg["n_leap_steps"], "step_size": cfg["step_size"],
                       "n_burnin_draws": cfg["n_burnin_draws"], "n_keep_draws": n_keep,
                       "parallelism": f"chains sharded x{world}, no data-path collective",
                       "accept_rate": acc_rate},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
                         "traffic": PROFILED_TRAFFIC_BYTES.get((C, d, cfg["n_leap_steps"], cfg["n_burnin_draws"], n_keep)),
                         "kernel": "hmc_gauss_mfma_kernel<8, 8>", "kernel_ms": k_ms,
                         "flop_per_unit": flop_per_unit},
        }
        out["ess_per_sec"] = ess_total_rank * world / (elapsed / args.steps)
        out["ess_note"] = ("min-over-dims Geyer-IPS ESS of the 100 kept draws (autocovariance pooled over all chains on the device), "
                           "x chains, / seconds per step")
        out["rhat_max"] = rhat_max
        if collate_ms is not None:
            out["collate_last_draw_allgather_ms"] = collate_ms
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, prec_h)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
