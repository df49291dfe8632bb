"""Launch shapes of the plain HMC kernel at the per-GPU chain counts of a strong-scaled BASELINE configs[1]
(65 536 chains over 1 / 2 / 4 / 8 GPUs): device-resident, HIP events, every shape at every count."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
dev = torch.device("cuda", 0)
d = 128
prec = torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=100, n_keep_draws=100, n_leap_steps=16, step_size=0.05)
shapes = [("auto", mcmc_amd.KERNEL_AUTO), ("2w/simd", mcmc_amd.KERNEL_HMC_TWO_WAVES_PER_SIMD), ("1w/simd", mcmc_amd.KERNEL_HMC_ONE_WAVE_PER_SIMD),
          ("split2", mcmc_amd.KERNEL_HMC_SPLIT2), ("split4x2", mcmc_amd.KERNEL_HMC_SPLIT4_TWO_WAVES), ("split4", mcmc_amd.KERNEL_HMC_SPLIT4)]
for C in [int(a) for a in sys.argv[1:]] or [65536, 32768, 16384, 8192, 4096]:
    theta0 = torch.from_numpy(np.ascontiguousarray(synth.initial_states(C, d, seed=3).T)).to(dev)
    theta = torch.empty_like(theta0)
    draws = torch.empty((100, d, C), dtype=torch.float64, device=dev)
    ref = None
    row = {"chains": C}
    for name, hint in shapes:
        t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        ch = mcmc_amd.make_chains(theta, C, draws=draws, mem=mcmc_amd.MEM_DEVICE)
        best = 1e9
        for rep in range(2):
            theta.copy_(theta0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); mcmc_amd.run("hmc", t, st, ch, stream=torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        if ref is None:
            ref = draws.clone()
        row[name] = round(best, 2)
        row[name + "_same"] = bool(torch.equal(ref, draws))
    print(json.dumps(row))
