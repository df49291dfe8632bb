"""CPU, world_size 2, gloo: chain sharding + draws collation of mcmc_amd.dist (the N>1 path).

The engine itself needs a GPU, so the per-shard runner here is the CPU oracle (tests may use it):
what is under test is the shard arithmetic, the global chain ids handed to the runner and the
all-gather layout."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import orc
from mcmc_amd import dist as mdist, synth
dist.init_process_group(backend="gloo")
d, C_total = 6, 13                      # 13 chains over 2 ranks: ragged shards 7 + 6
P = synth.dense_gaussian_precision(d, seed=4)

def runner(algo, kind, init, settings, chain0=0, **kw):
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P, W=4)
    return orc.run_many(orc.ALGO_HMC, t, init, settings, chain0=chain0)

st = orc.make_settings(seed=11, n_burnin=3, n_keep=5, n_leap=4, step=0.1, W=4)
init_fn = lambda chain0, c: synth.initial_states(c, d, seed=3, chain0=chain0)
draws, nacc = mdist.run_sharded("hmc", None, init_fn, C_total, st, runner=runner)
# more ranks than chains: rank 1's shard is empty, it must still join the collectives (and get the full result)
draws1, nacc1 = mdist.run_sharded("hmc", None, init_fn, 1, st, runner=runner)
assert draws1.shape == (5, d, 1) and nacc1.shape == (1,)
# overlapped collation: the kept draws in chunks chained through draw0, chunk k gathered while chunk k + 1 "samples".  The runner is a
# deterministic map of (global chain id, draw index), so any slip in the chunk bookkeeping (draw0, carried state, shard offsets) shows.
def fake(algo, kind, init, settings, chain0=0, draw0=0, **kw):
    x = np.array(init, dtype=np.float64, copy=True)
    b, k = int(settings.n_burnin_draws), int(settings.n_keep_draws)
    cid = chain0 + np.arange(x.shape[0])[:, None]
    rows, nacc_ = np.zeros((k, x.shape[1], x.shape[0])), np.zeros(x.shape[0], dtype=np.int64)
    for t in range(draw0, draw0 + b + k):
        x = 0.5 * x + np.sin(0.1 * cid + t + np.arange(x.shape[1])[None, :])
        if t - draw0 >= b:
            rows[t - draw0 - b] = x.T
            nacc_ += ((cid[:, 0] + t) % 3 == 0)
    return rows, dict(n_accept=nacc_, theta=x.T.copy(), eps=None)
st2 = orc.make_settings(seed=1, n_burnin=4, n_keep=11, n_leap=1, step=0.1, W=4)
ov_draws, ov_nacc = mdist.run_sharded_overlapped("hmc", None, init_fn, C_total, st2, 4, runner=fake)
one_draws, one_info = fake("hmc", None, init_fn(0, C_total), st2)
assert np.array_equal(ov_draws, one_draws) and np.array_equal(ov_nacc, one_info["n_accept"])
ov1, ov1n = mdist.run_sharded_overlapped("hmc", None, init_fn, 1, st2, 3, runner=fake)       # rank 1's shard is empty
assert np.array_equal(ov1, one_draws[:, :, :1]) and ov1n[0] == one_info["n_accept"][0]
if dist.get_rank() == 0:
    np.savez(sys.argv[2], draws=draws, nacc=nacc, draws1=draws1, nacc1=nacc1)
if dist.get_rank() == 1:
    np.savez(sys.argv[2] + ".rank1.npz", draws1=draws1)
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_all_chains_once():
    from mcmc_amd.dist import shard_bounds
    for total, world in [(65536, 8), (13, 2), (5, 8), (1 << 20, 8), (7, 3)]:
        got = []
        for r in range(world):
            c0, c = shard_bounds(total, world, r)
            got += list(range(c0, c0 + c))
        assert got == list(range(total))
    assert shard_bounds(65536, 8, 3) == (3 * 8192, 8192)


def test_two_rank_gloo_collation_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from mcmc_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), ROOT, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    got = np.load(out)
    d, C = 6, 13
    P = synth.dense_gaussian_precision(d, seed=4)
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=P, W=4)
    st = orc.make_settings(seed=11, n_burnin=3, n_keep=5, n_leap=4, step=0.1, W=4)
    want, info = orc.run_many(orc.ALGO_HMC, t, synth.initial_states(C, d, seed=3), st, chain0=0)
    assert got["draws"].shape == (5, d, C)
    assert np.array_equal(got["draws"], want)
    assert np.array_equal(got["nacc"], info["n_accept"].astype(np.int64))
    # world_size > n_chains: the one chain, identical on the rank that ran it and on the rank with the empty shard
    assert np.array_equal(got["draws1"], want[:, :, :1]) and got["nacc1"][0] == info["n_accept"][0]
    assert np.array_equal(np.load(str(out) + ".rank1.npz")["draws1"], want[:, :, :1])
