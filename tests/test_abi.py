"""CPU: the C-ABI library loads and exports every symbol include/mi_mcmc.h declares (no compute)."""
import ctypes
import os
import re

import pytest

import mcmc_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "mi_mcmc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for n in ("mi_mcmc_hmc_run", "mi_mcmc_mala_run", "mi_mcmc_nuts_run", "mi_settings_default",
              "mi_mcmc_last_error", "mi_mcmc_device_count"):
        assert n in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(mcmc_amd.LIB_PATH):
        pytest.skip("libmi_mcmc.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(mcmc_amd.LIB_PATH)
    for n in _declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/mi_mcmc.h but not exported"
    assert sorted(mcmc_amd.EXPORTS) == _declared_functions()


def test_probes_live_in_their_own_library_not_in_the_shipped_one():
    """mi_probe_* are test / measurement infrastructure (mcmc_amd/csrc/mi_mcmc_probes.h, libmi_mcmc_probes.so)."""
    if not os.path.exists(mcmc_amd.LIB_PATH):
        pytest.skip("libmi_mcmc.so not built")
    lib = ctypes.CDLL(mcmc_amd.LIB_PATH)
    plib = ctypes.CDLL(os.path.join(os.path.dirname(mcmc_amd.LIB_PATH), "libmi_mcmc_probes.so"))
    hdr = open(os.path.join(ROOT, "mcmc_amd", "csrc", "mi_mcmc_probes.h")).read()
    for n in mcmc_amd.PROBE_EXPORTS:
        assert not hasattr(lib, n), f"{n} exported from the shipped library"
        assert hasattr(plib, n) and n in hdr


def test_struct_sizes_and_defaults_match_reference_settings():
    if not os.path.exists(mcmc_amd.LIB_PATH):
        pytest.skip("libmi_mcmc.so not built")
    s = mcmc_amd.default_settings()
    # defaults of hmc_settings_t / nuts_settings_t (mcmc_structs.hpp:66-101)
    assert s.struct_size == ctypes.sizeof(mcmc_amd.mi_settings)
    assert (s.n_burnin_draws, s.n_keep_draws, s.n_leap_steps, s.step_size) == (1000, 1000, 1, 1.0)
    assert (s.n_adapt_draws, s.target_accept_rate, s.max_tree_depth) == (1000, 0.55, 10)
    assert (s.gamma_val, s.t0_val, s.kappa_val) == (0.05, 10.0, 0.75)
    assert s.vals_bound == 0 and not s.precond_mat
    assert mcmc_amd.lib().mi_mcmc_version() == 0x000600


def test_bad_arguments_are_rejected_without_a_gpu():
    if not os.path.exists(mcmc_amd.LIB_PATH):
        pytest.skip("libmi_mcmc.so not built")
    import numpy as np
    s = mcmc_amd.default_settings()
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_ISO, 4)
    c = mcmc_amd.make_chains(np.zeros((4, 2)), 2)
    t.struct_size = 3
    with pytest.raises(mcmc_amd.MiMcmcError) as e:
        mcmc_amd.run("hmc", t, s, c)
    assert e.value.code == mcmc_amd.MI_ERR_BAD_ARG
    t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_ISO, 4)
    if mcmc_amd.lib().mi_mcmc_device_count() == 0:
        with pytest.raises(mcmc_amd.MiMcmcError) as e:   # no CPU fallback: fails loudly
            mcmc_amd.run("hmc", t, s, c)
        assert e.value.code == mcmc_amd.MI_ERR_NO_DEVICE


def test_public_headers_stand_alone():
    """include/ is all a user-target build needs (the public target headers reached into mcmc_amd/csrc): the example target
    libraries and the C++ front-end parse and instantiate with a COPY of include/ alone on the include path."""
    import shutil
    import subprocess
    import tempfile
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        for ex in ("user_target.hip", "user_tile_target.hip"):
            shutil.copy(os.path.join(ROOT, "examples", ex), tmp)
            subprocess.check_call([hipcc, "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fsyntax-only", "-Wall", "-Wno-unused-function",
                                   "-Iinclude", ex], cwd=tmp)
        shutil.copy(os.path.join(ROOT, "examples", "hmc_plumbing.cpp"), tmp)
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Iinclude", "hmc_plumbing.cpp"], cwd=tmp)
