import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
