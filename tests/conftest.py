import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: the long slices of the GPU suite (full-width reference-order runs, the bigger fuzz sweeps); "
                                       "`-m gpu` alone leaves them out, `-m gpu_slow` (or any -m expression that names gpu_slow) runs them")


def pytest_collection_modifyitems(config, items):
    """The driver runs `pytest -m gpu` under a time limit (634 s of 1 200 and growing): tests marked gpu_slow are deselected
    unless the -m expression names gpu_slow itself.  Every gpu_slow test has a shorter sibling in the default suite (same code, fewer chains /
    cases), so no path is covered only by the slow set."""
    if "gpu_slow" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_slow") is not None else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    """The oracle is test infrastructure: (re)build it when a compiler is around."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "mcmc_oracle.c")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(os.path.join(ROOT, "oracle", f)) > os.path.getmtime(so)
        for f in ("mcmc_oracle.c", "mcmc_oracle.h", "orc_math.h"))
    if stale and os.path.exists(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    yield
