"""Timing of the NUTS kernels on BASELINE configs[3] (device-resident, HIP events), same box, alternating: MI_AB=memo_only (nuts_memo.hpp) |
memo (against the tick-local kernel, nuts_async.hpp: an independent implementation, same bits); MI_D = dimension; argv: chains [draws per half].
(The A/B runs against the retired nuts_reg / nuts_dyn / nuts_split kernels are in profiles/r5_nuts_ab_*.log.)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
burn = keep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
d = int(os.environ.get("MI_D", "128"))
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
theta = torch.empty_like(theta0)
draws = torch.empty((keep, d, C), dtype=torch.float64, device=dev)
n_leap = torch.zeros(C, dtype=torch.int64, device=dev)
n_exec = torch.zeros(C, dtype=torch.int64, device=dev)
skw = {}
if os.environ.get("MI_DIAGM") == "1":           # a diagonal precond_mat: nuts_gauss_memo_kernel<NT, true>
    skw["precond_mat"] = np.diag(np.linspace(0.5, 2.0, d))
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=burn, n_keep_draws=keep, n_adapt_draws=burn, **skw)
ref = None
for rep in range(2):
    AB = {"memo": (("memo", mcmc_amd.KERNEL_AUTO), ("tick_local", mcmc_amd.KERNEL_NUTS_TICK_LOCAL)),
          "memo_only": (("memo", mcmc_amd.KERNEL_AUTO),)}[os.environ.get("MI_AB", "memo_only")]
    for name, hint in AB:
        t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        ch = mcmc_amd.make_chains(theta, C, draws=draws, n_leapfrogs=n_leap, n_leapfrogs_executed=n_exec, mem=mcmc_amd.MEM_DEVICE)
        theta.copy_(theta0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); mcmc_amd.run("nuts", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        leaps = float(n_leap.double().sum().item())
        execd = float(n_exec.double().sum().item())
        chk = float(draws[-1].double().sum().item())
        if ref is None:
            ref = draws.clone()
        same = bool(torch.equal(ref, draws))
        print(json.dumps({"kernel": name, "chains": C, "ms": ms, "leapfrogs": leaps, "executed": execd, "lib_kernel": mcmc_amd.last_kernel(), "units_per_s": leaps * d / (ms * 1e-3),
                          "TFLOPs": leaps * d * 264 / (ms * 1e-3) / 1e12, "same_bits_as_first": same, "checksum": chk}))
