import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mcmc_amd
n, d, C = 100, 512, 65536
x = torch.randn((n, d, C), dtype=torch.float64, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    mcmc_amd.draw_stats(x, n, d, C, mem=mcmc_amd.MEM_DEVICE, want_acov=False)
    torch.cuda.synchronize(); print(os.environ.get("MI_MCMC_LIB", "default"), "%.1f ms for %.1f GB" % ((time.perf_counter() - t) * 1e3, x.numel() * 8 / 1e9))
