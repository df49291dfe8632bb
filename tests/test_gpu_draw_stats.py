"""GPU: device reducers over draws_out slabs (mi_mcmc_draw_stats) against the numpy definitions of mcmc_amd/ess.py."""
import numpy as np
import pytest

import mcmc_amd
from mcmc_amd.ess import ess_per_chain

pytestmark = pytest.mark.gpu


def _ar1(n, d, C, phi, seed):
    rng = np.random.default_rng(seed)
    y = np.zeros((n, d, C))
    y[0] = rng.standard_normal((d, C))
    for t in range(1, n):
        y[t] = phi * y[t - 1] + np.sqrt(1 - phi ** 2) * rng.standard_normal((d, C))
    return y + np.arange(d)[None, :, None]          # a different mean per dimension


@pytest.mark.parametrize("n,d,C,phi", [(100, 5, 700, 0.6), (40, 3, 64, 0.0), (160, 2, 1000, 0.9), (7, 4, 130, 0.3)])
def test_draw_stats_match_the_numpy_definitions(n, d, C, phi):
    x = _ar1(n, d, C, phi, seed=n)
    s = mcmc_amd.draw_stats(x)
    assert np.allclose(s["mean"], x.mean(axis=(0, 2)), rtol=1e-12, atol=1e-12)
    xc = x - x.mean(axis=(0, 2), keepdims=True)
    acov = np.stack([(xc[: n - k] * xc[k:]).sum(axis=0).mean(axis=1) / (n - k) for k in range(n)])
    assert np.allclose(s["acov"], acov, rtol=1e-10, atol=1e-12)
    assert np.allclose(s["ess"], ess_per_chain(x), rtol=1e-8)
    m = x.mean(axis=0); W = x.var(axis=0, ddof=1).mean(axis=1); B_n = m.var(axis=1, ddof=1)
    assert np.allclose(s["rhat"], np.sqrt(((n - 1) / n * W + B_n) / W), rtol=1e-10)
    assert np.all(np.abs(s["rhat"] - 1.0) < 0.2)


def test_draw_stats_of_a_real_run():
    from mcmc_amd import synth
    d, C, keep = 16, 4096, 60
    prec = synth.dense_gaussian_precision(d)
    st = mcmc_amd.default_settings(rng_seed_value=1, n_burnin_draws=50, n_keep_draws=keep, n_leap_steps=8, step_size=0.15)
    draws, _ = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, synth.initial_states(C, d, seed=3), st, prec=prec)
    s = mcmc_amd.draw_stats(draws)
    assert np.allclose(s["ess"], ess_per_chain(draws), rtol=1e-8)
    assert np.all(s["rhat"] < 1.05) and np.all(np.abs(s["mean"]) < 0.1)


@pytest.mark.parametrize("n,d,C,phi", [(1000, 3, 200, 0.7), (161, 2, 130, 0.2), (517, 2, 64, 0.95)])
def test_long_series_are_streamed_and_truncated_at_128_lags(n, d, C, phi):
    """n_keep beyond 160 (the reference's default is 1 000 kept draws): lags 0..127 from the time-tiled kernel, NaN beyond;
    mean, R-hat and Geyer's ESS (which stops at the first non-positive pair, long before lag 127 here) as defined in ess.py."""
    x = _ar1(n, d, C, phi, seed=n)
    s = mcmc_amd.draw_stats(x)
    assert np.allclose(s["mean"], x.mean(axis=(0, 2)), rtol=1e-12, atol=1e-12)
    xc = x - x.mean(axis=(0, 2), keepdims=True)
    acov = np.stack([(xc[: n - k] * xc[k:]).sum(axis=0).mean(axis=1) / (n - k) for k in range(128)])
    assert np.allclose(s["acov"][:128], acov, rtol=1e-10, atol=1e-12) and np.isnan(s["acov"][128:]).all()
    m = x.mean(axis=0); W = x.var(axis=0, ddof=1).mean(axis=1); B_n = m.var(axis=1, ddof=1)
    assert np.allclose(s["rhat"], np.sqrt(((n - 1) / n * W + B_n) / W), rtol=1e-10)
    if phi < 0.9:
        assert np.allclose(s["ess"], ess_per_chain(x), rtol=1e-8)
    else:                                   # the sum may still be positive at lag 127: the device reports the truncated sum
        assert np.all(s["ess"] >= ess_per_chain(x) * (1 - 1e-8)) and np.all(s["ess"] < n)


@pytest.mark.parametrize("n,d,C,phi", [(100, 5, 700, 0.6), (40, 3, 64, 0.0), (160, 2, 1000, 0.9), (7, 4, 130, 0.3), (1000, 3, 200, 0.7),
                                       (100, 2, 300, 0.97), (20, 130, 70, 0.5)])
def test_ess_and_rhat_without_the_full_autocovariance(n, d, C, phi):
    """acov == NULL: lags in blocks of 16, then 32, straight from HBM (stats_window_kernel), until Geyer's sum has ended in every
    dimension; phi = 0.97 does not end within 32 lags and falls through to the full computation.  Same numbers as ess.py."""
    x = _ar1(n, d, C, phi, seed=n + C)
    s = mcmc_amd.draw_stats(x, want_acov=False)
    assert s["acov"] is None
    assert np.allclose(s["mean"], x.mean(axis=(0, 2)), rtol=1e-12, atol=1e-12)
    if n <= 160:
        assert np.allclose(s["ess"], ess_per_chain(x), rtol=1e-8)
    else:
        full = mcmc_amd.draw_stats(x)["ess"]
        assert np.allclose(s["ess"], full, rtol=1e-8)
    m = x.mean(axis=0); W = x.var(axis=0, ddof=1).mean(axis=1); B_n = m.var(axis=1, ddof=1)
    assert np.allclose(s["rhat"], np.sqrt(((n - 1) / n * W + B_n) / W), rtol=1e-10)


def test_device_transpose_to_chain_major():
    import torch
    n, d, C = 37, 5, 1000
    x = np.random.default_rng(0).standard_normal((n, d, C))
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((C, d, n), dtype=torch.float64, device="cuda")
    mcmc_amd.draws_to_chain_major_device(xd, n, d, C, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.ascontiguousarray(x.transpose(2, 1, 0)))
    n, d, C = 300, 2, 130                                   # more draws than one LDS tile holds
    x = np.random.default_rng(1).standard_normal((n, d, C))
    out = torch.empty((C, d, n), dtype=torch.float64, device="cuda")
    mcmc_amd.draws_to_chain_major_device(torch.from_numpy(x).cuda(), n, d, C, out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.ascontiguousarray(x.transpose(2, 1, 0)))
