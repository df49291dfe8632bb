"""GPU: randomised parity sweep (tests/fuzz_parity.py) -- every device sampler against the oracle on random small cases:
three samplers x four targets x bounds / diagonal precond / degenerate sizes / step sizes that blow the chain up."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


SLOW = pytest.mark.gpu_slow      # the second / larger slice of a sweep: -m gpu_slow (the first slice stays in the default suite)


@pytest.mark.parametrize("seed", [7, pytest.param(11, marks=SLOW)])
def test_random_cases_are_bit_exact(seed):
    import fuzz_parity
    assert fuzz_parity.sweep(60, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [3, pytest.param(5, marks=SLOW)])
def test_random_normal_model_cases_are_bit_exact(seed):
    """hmc / mala / rwmh / rmhmc / nuts on the one-lane-per-chain engine (d = 2 normal model)"""
    import fuzz_parity
    assert fuzz_parity.sweep_small(80, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [5])
def test_plain_nuts_kernel_random_windows_and_depth_caps(seed):
    """Draw boundaries of the plain NUTS kernels (momenta ahead, epilogue on the spot, deferred rows): random burn-in / keep / adaptation
    windows, depth caps 0..10, chain counts around the wave size."""
    import fuzz_nuts
    assert fuzz_nuts.sweep(40, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [7])
def test_nuts_on_the_lds_streamed_evaluation_random_cases(seed):
    """nuts_lds.hpp against literal_kernel<2> of the same library (both on the GPU): random targets of every instantiation, ragged
    workgroups, windows, depth caps, step sizes, a diagonal precond_mat, non-finite starts, runs cut in two."""
    import fuzz_nuts_lds
    assert fuzz_nuts_lds.sweep(16, seed, verbose=False) == 0


@pytest.mark.parametrize("n", [6, pytest.param(16, marks=SLOW)])
def test_per_chain_mass_sweep_is_bit_exact(n):
    """hmc with mi_chains.mass_diag: chain c against the oracle with precond_mat = diag(mass[:, c]), random targets / sizes / bounds /
    non-finite starts (elementwise and literal kernels)"""
    import fuzz_parity
    assert fuzz_parity.sweep_mass(n, 3, verbose=False) == 0


@pytest.mark.parametrize("n", [40, pytest.param(120, marks=SLOW)])
def test_general_variants_with_chains_started_non_finite(n):
    """bounds / diagonal / dense precond_mat x dense / diag / iso targets x dimensions that do not fill their tiles x chains that start at
    +-inf / NaN / 1e300 (tests/fuzz_nonfinite_general.py; round 5: caught mala_gauss_dense_m_kernel on ISO / DIAG targets)"""
    import fuzz_nonfinite_general
    assert fuzz_nonfinite_general.sweep(n, 5, verbose=False) == 0


def test_hmc_with_a_dense_precond_mat_on_the_streamed_kernels_random_cases():
    """logit_lds_kernel<.., DENSEM> (round 5): random sizes of both targets, 0..5 leapfrog steps, the non-finite regime"""
    import fuzz_parity
    assert fuzz_parity.sweep_dense_m(16, 3, verbose=False) == 0


@pytest.mark.parametrize("n", [12, pytest.param(60, marks=SLOW)])
def test_matrix_product_samplers_random_cases(n):
    """gemm_samplers.hip (hmc / mala / rwmh beyond d = 512, dense Gaussians and the logistic target) against the literal kernels of the same library:
    ragged d / N / chain tiles, step sizes from tiny to absurd, non-finite starts, chain0 / draw0 offsets, runs cut in two (tests/fuzz_gemm.py)"""
    import fuzz_gemm
    assert fuzz_gemm.sweep(n, 9, verbose=False) == 0
