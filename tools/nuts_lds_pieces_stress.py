"""Stress of the piece hand-over of nuts_lds.hpp (runs cut into pieces that migrate between chain slots -- and XCDs -- through memory inside ONE launch):
a few tens of thousands of chains, some started non-finite, short runs repeated; every repetition must reproduce the bits of the SAME run made in shards of at
most 4 096 chains (fewer chains than chain slots: whole chains in fixed slots, no pieces; global chain ids through mi_chains.chain0).
python tools/nuts_lds_pieces_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, mcmc_amd
from mcmc_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
eq = lambda a, b: bool(torch.equal(a.view(torch.int64), b.view(torch.int64)) or torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0)))
bad = 0
for kind, d, n_rows, C, half, depth in [("logistic", 512, 64, 32768, 8, 6), ("dense", 256, 0, 65536, 10, 7), ("logistic", 100, 128, 65536, 4, 6)]:
    init = synth.initial_states(C, d, seed=3) * (0.1 if kind == "logistic" else 1.0)
    init[5] = 1e300; init[C // 3, 7] = np.inf; init[C // 2, d - 1] = np.nan; init[C - 1] = 1e160
    theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    if kind == "dense":
        tkw = dict(prec=torch.from_numpy(synth.dense_gaussian_precision(d)).to(dev)); tk = mcmc_amd.TARGET_GAUSS_DENSE
    else:
        X, y = synth.logistic_problem(d, n_rows, seed=1)
        tkw = dict(X=torch.from_numpy(X).to(dev), y=torch.from_numpy(y).to(dev)); tk = mcmc_amd.TARGET_LOGISTIC
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=half, n_keep_draws=half, n_adapt_draws=half, max_tree_depth=depth)
    tgt = mcmc_amd.make_target(tk, d, mem=mcmc_amd.MEM_DEVICE, **tkw)
    def run(c0, c1):
        n = c1 - c0
        theta = theta0[:, c0:c1].clone().contiguous()         # (a copy: the call writes the final states back)
        draws = torch.empty((half, d, n), dtype=torch.float64, device=dev)
        n_leap = torch.zeros(n, dtype=torch.int64, device=dev); eps = torch.zeros(n, dtype=torch.float64, device=dev); nacc = torch.zeros(n, dtype=torch.int64, device=dev)
        ch = mcmc_amd.make_chains(theta, n, draws=draws, n_leapfrogs=n_leap, step_size=eps, n_accept=nacc, mem=mcmc_amd.MEM_DEVICE, chain0=c0)
        mcmc_amd.run("nuts", tgt, st, ch, stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        return draws, theta, n_leap, eps, nacc
    shard = 4096
    parts = [run(c0, c0 + shard) for c0 in range(0, C, shard)]
    ref = [torch.cat([p[0] for p in parts], dim=2), torch.cat([p[1] for p in parts], dim=1)] + [torch.cat([p[k] for p in parts]) for k in (2, 3, 4)]
    print(kind, d, C, "reference in shards of", shard, ":", mcmc_amd.last_kernel(), "leapfrogs", int(ref[2].sum().item()), flush=True)
    for r in range(reps):
        got = run(0, C)
        ok = eq(got[0], ref[0]) and eq(got[1], ref[1]) and bool(torch.equal(got[2], ref[2])) and eq(got[3], ref[3]) and bool(torch.equal(got[4], ref[4]))
        bad += 0 if ok else 1
        print(("ok  " if ok else "FAIL"), r, mcmc_amd.last_kernel(), flush=True)
print("mismatching repetitions:", bad)
sys.exit(1 if bad else 0)
