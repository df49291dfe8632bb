"""CPU: the host logic of bench.py that needs no GPU -- which committed profile a traffic figure may be quoted from, and that an N-GPU
line is refused when the devices are not there."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_is_only_quoted_from_a_profile_of_the_kernel_that_ran():
    """profiles/ lookup: newest round first, and only a PMC pass of this workload AND of the kernel mi_mcmc_last_kernel named."""
    import bench
    key = (4, 65536, 128, 200)
    t, src = bench.profiled_traffic(4, key, "nuts_gauss_dyn_kernel<8, false>")
    assert t is not None and src.startswith("profiles/r") and src.endswith("_c4_pmc.json")
    assert "nuts_gauss_dyn_kernel<8, false>" in json.load(open(os.path.join(ROOT, src)))["derived"]["kernel"]
    t2, why = bench.profiled_traffic(4, key, "nuts_gauss_no_such_kernel<8>")
    assert t2 is None and "not quoted" in why
    tm, srcm = bench.profiled_traffic(4, key, "nuts_gauss_memo_kernel<8, false, true>")    # what configs[3] runs on since round 6 (momenta from the pre-pass table)
    assert tm is not None and tm < t and "nuts_gauss_memo_kernel<8, false, true>" in json.load(open(os.path.join(ROOT, srcm)))["derived"]["kernel"]
    ti, whyi = bench.profiled_traffic(4, key, "nuts_gauss_memo_kernel<8, false, false>")   # the in-tick instantiation was never profiled under this name: no figure is made up
    assert ti is None and "not quoted" in whyi
    t3, why3 = bench.profiled_traffic(4, (4, 1234, 128, 200), "nuts_gauss_dyn_kernel<8, false>")
    assert t3 is None and "another workload shape" in why3


def test_gpus_flag_refuses_when_the_devices_are_not_there():
    """`bench.py --gpus 2` on a box with fewer than 2 visible GPUs: rc != 0 and no JSON line (it used to run one rank and print n_gpus 1)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs are visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_TEST_SHARE_GPU")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in out.stderr


def test_gpus_flag_must_match_the_launcher():
    """Under a launcher (WORLD_SIZE set) a mismatching --gpus is an error, not a note on stderr."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
