"""GPU: bench.py's output contract, at N=1 and through the N=2 code path (two ranks sharing GPU 0 over gloo: BENCH_TEST_SHARE_GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--chains", "8192", "--no-cpu-baseline", "--traffic", "none"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 2 and j["vs_baseline"] is None and j["dtype"] == "f64"
    assert j["roofline"]["bound"] == "mfma" and 0.0 < j["roofline"]["frac"] < 1.0 and "workload" in j["config"]
    # the kernel label is what the engine launched (mi_mcmc_last_kernel), not a static table: 8 192 chains take a split-tile shape
    assert j["roofline"]["kernel"].startswith("hmc_gauss_split_kernel<8, 4,"), j["roofline"]["kernel"]
    assert "ess_per_sec" in j and j["ess_per_sec_incl_reducer"] < j["ess_per_sec"]


def test_traffic_is_measured_by_rocprofv3_child_runs():
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    out = subprocess.run([sys.executable, "bench.py", "--config", "5", "--steps", "1", "--warmup", "0", "--chains", "4096",
                          "--no-cpu-baseline", "--no-ess", "--traffic", "all"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out.stdout)["roofline"]
    assert r["kernel"].startswith("hmc_diag4_kernel")
    assert isinstance(r["traffic_source"], dict) and r["traffic_source"]["source"].startswith("rocprofv3"), r["traffic_source"]
    # 4096 chains x 1024 dims x 8 B: every draw reads the state and writes the proposal once -- 28 draws, between 1x and 4x of that
    algorithmic = 28 * 2 * 4096 * 1024 * 8
    assert 0.5 * algorithmic < r["traffic"] < 4.0 * algorithmic, (r["traffic"], algorithmic)


def test_two_rank_code_path():
    env = dict(os.environ, BENCH_TEST_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "8193",
           "--no-cpu-baseline", "--traffic", "none"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    # north_star: strong scaling -- with WORLD_SIZE > 1 the total chain count is fixed and sharded (8193: ragged, nothing dropped)
    assert j["n_gpus"] == 2 and j["config"]["chains_total"] == 8193 and j["config"]["chains_per_gpu"] == 4097 and j["scaling"] == "strong"
    assert j["value"] > 0
    _collation_on_the_line(j, shared_device=True)


def _collation_on_the_line(j, shared_device):
    """a line of more than one rank ALWAYS carries the collation of draws_out, through the C ABI: `value` (sampling only),
    `value_incl_collation` (then one blocking all-gather), `value_overlapped` (4 chunks, _begin / _wait), `collate_GBps`, the RCCL rank count."""
    assert 0 < j["value_incl_collation"] < j["value"]
    assert j["value_overlapped"] > 0 and j["collate_GBps"] > 0
    c = j["collation"]
    assert "mi_mcmc_allgather_draws_rank_major" in c["abi"] and "mi_mcmc_allgather_draws_begin" in c["abi"]
    assert c["blocking_ms"] > 0 and c["overlapped_total_ms"] > 0 and c["overlapped_chunks"] == 4
    assert c["own_shard_in_place"] is True and c["overlapped_equals_blocking_run"] is True
    if shared_device:          # two ranks on ONE device: RCCL refuses them in one communicator, each rank collates over its own
        assert c["rccl_ranks"] == 1 and "NOT a multi-GPU collation" in c["note"]
    else:
        assert c["rccl_ranks"] == j["n_gpus"]


def test_gpus_flag_alone_launches_the_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (what a plain driver command would be): bench.py spawns the two ranks itself and the
    line says n_gpus 2 with the rank count the process group saw (it used to run one rank and print n_gpus 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_TEST_SHARE_GPU"] = "1"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "8192", "--no-cpu-baseline",
                          "--traffic", "none"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert j["n_gpus"] == 2 and j["ranks"]["world_size"] == 2 and j["config"]["chains_per_gpu"] == 4096 and j["scaling"] == "strong"
    _collation_on_the_line(j, shared_device=True)
    # rank 0's roofline quotes the committed counter pass of ONE GPU'S SHARE when there is one of this kernel (traffic is not null at N > 1)
    assert "traffic" in j["roofline"]


def test_gpus_flag_refuses_more_ranks_than_devices():
    """Without the test-only sharing switch an N-GPU line needs N visible GPUs: rc != 0, no JSON line."""
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_TEST_SHARE_GPU")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in out.stderr


def test_collate_flag_runs_the_c_abi_collation_on_one_rank():
    """--collate at one GPU: the same C-ABI calls over a one-rank communicator (the code path, not a transfer)."""
    out = subprocess.run([sys.executable, "bench.py", "--collate", "--steps", "1", "--warmup", "0", "--chains", "4096", "--no-cpu-baseline",
                          "--traffic", "none", "--no-ess"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert j["collation"]["bytes_received_per_rank"] == 100 * 128 * 4096 * 8 and j["collation"]["rccl_ranks"] == 1
    _collation_on_the_line(j, shared_device=False)


@pytest.mark.parametrize("config,bound", [(3, "mfma"), (4, "mfma"), (5, "valu-fp64")])
def test_other_baseline_configs_print_their_own_roofline(config, bound):
    out = subprocess.run([sys.executable, "bench.py", "--config", str(config), "--steps", "1", "--warmup", "0", "--chains", "2048",
                          "--no-cpu-baseline", "--traffic", "none"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _line(out.stdout)
    assert KEYS <= set(j) and j["roofline"]["bound"] == bound and 0.0 < j["roofline"]["frac"] < 1.0
    assert j["roofline"]["traffic"] is None          # not the profiled workload: no traffic figure is made up
    assert j["config"]["chains_per_gpu"] == 2048
