"""GPU: INV / CHOL_LOWER of a dense precond_mat on the device (mcmc_amd/csrc/linalg_device.hip through mi_mcmc_mat_inverse /
mi_mcmc_mat_cholesky_lower) against the oracle's orc_inv / orc_chol_lower (ref: src/hmc.cpp:58-59, src/mala.cpp:58,
include/stats/dmvnorm.hpp:36-41): every element bit for bit -- random SPD matrices, ill-conditioned ones, matrices whose elimination has to
pivot (also on ties and across NaN), singular / non-SPD input (inf / NaN out, as the reference's unvalidated calls give), sizes around the
64-dimension switch between the calling thread's loops and the device, and sizes that do not divide the grid."""
import ctypes as C
import time

import numpy as np
import pytest

import mcmc_amd
import orc


def _orc_inv(A):
    d = A.shape[0]
    out = np.empty((d, d))
    orc.lib().orc_inv(orc._p(np.ascontiguousarray(A)), C.c_size_t(d), orc._p(out))
    return out


def _orc_chol(A):
    d = A.shape[0]
    out = np.empty((d, d))
    orc.lib().orc_chol_lower(orc._p(np.ascontiguousarray(A)), C.c_size_t(d), orc._p(out))
    return out


def _spd(d, seed, cond=None):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    M = A @ A.T + np.diag(rng.uniform(0.3, 3.0, d))
    if cond is not None:                     # D M D with D log-spaced: still SPD, condition number ~ cond^2 x M's
        D = np.logspace(0.0, -np.log10(cond), d)
        rng.shuffle(D)
        M = D[:, None] * M * D[None, :]
    return M


def _same(a, b):
    """bit for bit, signed zeros included; NaN equals NaN (its sign / payload is the hardware's, x86 and gfx950 differ, and nothing reads it)"""
    fin = ~np.isnan(a)
    return np.array_equal(a, b, equal_nan=True) and np.array_equal(np.signbit(a[fin]), np.signbit(b[fin]))


@pytest.mark.gpu
@pytest.mark.parametrize("d", [64, 65, 100, 128, 255, 256, 300, 512])
def test_device_inverse_and_cholesky_equal_the_oracle_bitwise_on_spd_matrices(d):
    M = _spd(d, d)
    assert _same(mcmc_amd.mat_inverse(M), _orc_inv(M))
    assert _same(mcmc_amd.mat_cholesky_lower(M), _orc_chol(M))
    assert np.abs(mcmc_amd.mat_inverse(M) @ M - np.eye(d)).max() < 1e-8          # ... and it IS the inverse


@pytest.mark.gpu
@pytest.mark.parametrize("d,cond", [(96, 1e3), (200, 1e5), (512, 1e4)])
def test_ill_conditioned_matrices_whose_elimination_pivots(d, cond):
    M = _spd(d, 7 * d, cond)
    assert (np.abs(M).argmax(axis=0) != np.arange(d)).any()    # columns whose diagonal entry is not the largest: rows swap
    assert _same(mcmc_amd.mat_inverse(M), _orc_inv(M))
    assert _same(mcmc_amd.mat_cholesky_lower(M), _orc_chol(M))


@pytest.mark.gpu
def test_general_matrices_pivot_ties_nan_and_singular_input():
    rng = np.random.default_rng(5)
    d = 130
    G = rng.standard_normal((d, d))                             # not symmetric: INV does not care, pivots everywhere
    assert _same(mcmc_amd.mat_inverse(G), _orc_inv(G))
    T = np.round(rng.standard_normal((d, d)) * 2.0)             # small integers: exact ties in the pivot search (the FIRST largest wins), zeros skipped
    T += np.eye(d) * 3
    assert _same(mcmc_amd.mat_inverse(T), _orc_inv(T))
    N = _spd(d, 11)
    N[40, 3] = np.nan                                           # a NaN below the diagonal never wins the pivot search, then poisons what it touches
    assert _same(mcmc_amd.mat_inverse(N), _orc_inv(N))
    N2 = _spd(d, 12)
    N2[5, 5] = np.nan                                           # a NaN ON the diagonal stays the pivot
    assert _same(mcmc_amd.mat_inverse(N2), _orc_inv(N2))
    S = _spd(d, 13)
    S[:, 9] = 0.0; S[9, :] = 0.0                                # singular: division by a zero pivot, inf / NaN entries, no error
    assert _same(mcmc_amd.mat_inverse(S), _orc_inv(S))
    Q = -_spd(d, 14)                                            # not positive definite: sqrt of a negative number
    assert _same(mcmc_amd.mat_cholesky_lower(Q), _orc_chol(Q))
    assert np.isnan(mcmc_amd.mat_cholesky_lower(Q)).any()


@pytest.mark.gpu
def test_small_matrices_stay_on_the_calling_thread_and_agree_too():
    for d in (1, 2, 7, 63):
        M = _spd(d, 100 + d)
        assert _same(mcmc_amd.mat_inverse(M), _orc_inv(M)) and _same(mcmc_amd.mat_cholesky_lower(M), _orc_chol(M))


@pytest.mark.gpu
def test_device_factorisation_time_at_d512():
    """the host's Gauss-Jordan was 89 ms at d = 512 inside every blocking call with a dense precond_mat."""
    M = _spd(512, 3)
    mcmc_amd.mat_inverse(_spd(512, 4)); mcmc_amd.mat_cholesky_lower(_spd(512, 4))      # code-object load (another matrix: M is not memoised)
    t0 = time.perf_counter(); mcmc_amd.mat_inverse(M); t_inv = time.perf_counter() - t0
    t0 = time.perf_counter(); mcmc_amd.mat_cholesky_lower(M); t_chol = time.perf_counter() - t0
    print(f"d = 512: INV {t_inv * 1e3:.1f} ms, CHOL_LOWER {t_chol * 1e3:.1f} ms (device, incl. upload / download)")
    assert t_inv < 0.040 and t_chol < 0.040
