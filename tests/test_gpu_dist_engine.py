"""GPU, world_size 2: mcmc_amd.dist.run_sharded with the ENGINE as the per-shard runner (the product's N > 1 path).

Two ranks share GPU 0 (a 1-GPU box; bind_device maps LOCAL_RANK onto the visible devices) over gloo.  Every rank samples
its shard device-resident with global chain ids; the collated draws / n_accept must be bit-identical to ONE engine call
over all chains.  Shapes: BASELINE configs[1] (d = 128 dense Gaussian, L = 16) and one GPU's shard shape of configs[4]
(d = 1024 ill-conditioned diagonal Gaussian, L = 32), ragged shard sizes, and world_size > n_chains (an empty shard)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import mcmc_amd
from mcmc_amd import dist as mdist, synth
dist.init_process_group(backend="gloo")
out = {}
for name, kind, d, C_total, L, eps, burn, keep, cond in [
        ("c2", mcmc_amd.TARGET_GAUSS_DENSE, 128, 4099, 16, 0.05, 6, 10, None),      # ragged: 2050 + 2049
        ("c5", mcmc_amd.TARGET_GAUSS_DIAG, 1024, 1001, 32, 0.005, 2, 3, 1.0e4),      # ragged: 501 + 500
        ("one", mcmc_amd.TARGET_GAUSS_DENSE, 128, 1, 16, 0.05, 2, 4, None)]:         # rank 1 has no chain
    prec = synth.dense_gaussian_precision(d) if cond is None else synth.ill_conditioned_diag(d, cond)
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
    init_fn = lambda chain0, c, d=d: synth.initial_states(c, d, seed=3, chain0=chain0)
    draws, nacc = mdist.run_sharded("hmc", kind, init_fn, C_total, st, prec=prec)
    assert draws.is_cuda and nacc.is_cuda and draws.shape == (keep, d, C_total)
    out[name + "_draws"] = draws.cpu().numpy(); out[name + "_nacc"] = nacc.cpu().numpy()
# overlapped collation (run_sharded_overlapped): kept draws in chunks chained through draw0, chunk k all-gathered while chunk k + 1 samples
prec = synth.dense_gaussian_precision(128)
init_fn = lambda chain0, c: synth.initial_states(c, 128, seed=3, chain0=chain0)
st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=6, n_keep_draws=10, n_leap_steps=16, step_size=0.05)
draws, nacc = mdist.run_sharded_overlapped("hmc", mcmc_amd.TARGET_GAUSS_DENSE, init_fn, 4099, st, 3, prec=prec)
assert draws.is_cuda and draws.shape == (10, 128, 4099)
out["ov_hmc_draws"] = draws.cpu().numpy(); out["ov_hmc_nacc"] = nacc.cpu().numpy()
stn = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=8, n_keep_draws=9, n_adapt_draws=8, max_tree_depth=5, step_size=0.1)
draws, nacc = mdist.run_sharded_overlapped("nuts", mcmc_amd.TARGET_GAUSS_DENSE, init_fn, 333, stn, 4, prec=prec)
out["ov_nuts_draws"] = draws.cpu().numpy(); out["ov_nuts_nacc"] = nacc.cpu().numpy()
np.savez(sys.argv[2] + f".rank{dist.get_rank()}.npz", **out)
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_engine_shards_collate_to_the_single_call_result_bitwise(tmp_path):
    import mcmc_amd
    from mcmc_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), ROOT, out]
    r = subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    got = [np.load(out + f".rank{k}.npz") for k in (0, 1)]
    for name, kind, d, C_total, L, eps, burn, keep, cond in [
            ("c2", mcmc_amd.TARGET_GAUSS_DENSE, 128, 4099, 16, 0.05, 6, 10, None),
            ("c5", mcmc_amd.TARGET_GAUSS_DIAG, 1024, 1001, 32, 0.005, 2, 3, 1.0e4),
            ("one", mcmc_amd.TARGET_GAUSS_DENSE, 128, 1, 16, 0.05, 2, 4, None)]:
        prec = synth.dense_gaussian_precision(d) if cond is None else synth.ill_conditioned_diag(d, cond)
        st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=burn, n_keep_draws=keep, n_leap_steps=L, step_size=eps)
        want, info = mcmc_amd.hmc(kind, synth.initial_states(C_total, d, seed=3), st, prec=prec)      # ONE call, all chains
        for g in got:                                                                               # every rank holds the full result
            assert np.array_equal(g[name + "_draws"], want), name
            assert np.array_equal(g[name + "_nacc"], info["n_accept"].astype(np.int64)), name
        assert info["n_accept"].sum() > 0

    # the overlapped, chunked collation gives the same bytes as ONE call over all chains
    prec = synth.dense_gaussian_precision(128)
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=6, n_keep_draws=10, n_leap_steps=16, step_size=0.05)
    want, info = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, synth.initial_states(4099, 128, seed=3), st, prec=prec)
    stn = mcmc_amd.default_settings(rng_seed_value=7, n_burnin_draws=8, n_keep_draws=9, n_adapt_draws=8, max_tree_depth=5, step_size=0.1)
    wantn, infon = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, synth.initial_states(333, 128, seed=3), stn, prec=prec)
    for k in (0, 1):
        assert np.array_equal(got[k]["ov_hmc_draws"], want) and np.array_equal(got[k]["ov_hmc_nacc"], info["n_accept"].astype(np.int64))
        assert np.array_equal(got[k]["ov_nuts_draws"], wantn) and np.array_equal(got[k]["ov_nuts_nacc"], infon["n_accept"].astype(np.int64))
