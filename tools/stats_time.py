"""Time mi_mcmc_draw_stats on a configs[1]-shaped slab of synthetic AR(1) draws (GPU box):  python tools/stats_time.py [phi]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, mcmc_amd
phi = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
n, d, C = 100, 128, 65536
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.empty((n, d, C), dtype=torch.float64, device="cuda")
x[0] = torch.randn((d, C), dtype=torch.float64, device="cuda", generator=g)
for t in range(1, n):
    x[t] = phi * x[t - 1] + (1 - phi * phi) ** 0.5 * torch.randn((d, C), dtype=torch.float64, device="cuda", generator=g)
st = torch.cuda.current_stream().cuda_stream
for want in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = mcmc_amd.draw_stats(x, n, d, C, mem=mcmc_amd.MEM_DEVICE, stream=st, want_acov=want)
    torch.cuda.synchronize()
    print(f"phi={phi} want_acov={want}: {(time.perf_counter() - t0) * 1e3:.2f} ms  ess_min={s['ess'].min():.2f}")
