"""GPU: BASELINE.json configs 3, 4, 5 at their full sizes -- the oracle re-runs a sample of the chains (by
global chain id) bit-exactly; the population is checked through properties the domain offers."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def test_config3_mala_logistic_d512_262144_chains():
    d, N, C = 512, 1024, 262144
    X, y = synth.logistic_problem(d, N)
    init = np.zeros((C, d))
    init[:, 0] = np.linspace(-0.5, 0.5, C)                 # distinct starts
    st = mcmc_amd.default_settings(rng_seed_value=6, n_burnin_draws=3, n_keep_draws=2, step_size=0.02)
    draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y)
    assert draws.shape == (2, d, C) and np.isfinite(draws).all()
    pick = [0, 17, 4095, 131072, 262143]
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=4, block_size=128, eta_chains=2)
    for c in pick:
        s = orc.make_settings(seed=6, n_burnin=3, n_keep=2, step=0.02, W=4, hoist=1, blocks=4, block_size=128, chain_id=c)
        o, info = orc.run_chain(orc.ALGO_MALA, t, init[c], s)
        assert np.array_equal(draws[:, :, c], o) and g["n_accept"][c] == info["n_accept"]
    assert g["n_accept"].mean() / 2 > 0.9                   # eps = 0.02 at d = 512: nearly every move accepted


def test_config4_nuts_d128_65536_chains():
    d, C = 128, 65536
    prec = synth.dense_gaussian_precision(d)
    init = synth.initial_states(C, d, seed=3)
    st = mcmc_amd.default_settings(rng_seed_value=4, n_burnin_draws=6, n_keep_draws=3, n_adapt_draws=6, max_tree_depth=10)
    draws, g = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert draws.shape == (3, d, C) and np.isfinite(draws).all()
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec, W=4)
    for c in [0, 15, 16, 40000, 65535]:
        s = orc.make_settings(seed=4, n_burnin=6, n_keep=3, n_adapt=6, step=1.0, W=4, chain_id=c)
        o, info = orc.run_chain(orc.ALGO_NUTS, t, init[c], s, traces=True)
        assert np.array_equal(draws[:, :, c], o)
        assert np.array_equal(g["depth"][:, c], info["depth"]) and g["n_leap"][c] == info["n_leap"]
        assert g["eps"][c] == info["eps"]
    assert g["depth"].max() <= 10 and g["n_leap"].min() >= 9


def test_config5_hmc_d1024_diag_one_gpu_shard():
    d, C = 1024, 131072                                     # one GPU's share of 2^20 chains
    prec = synth.ill_conditioned_diag(d, 1.0e4)
    chain0 = 3 * C                                          # as rank 3 of 8 would run it
    init = (synth.initial_states(C, d, seed=3, chain0=chain0) / np.sqrt(prec)[None, :])
    st = mcmc_amd.default_settings(rng_seed_value=8, n_burnin_draws=2, n_keep_draws=2, n_leap_steps=32, step_size=0.005)
    draws, g = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DIAG, init, st, prec=prec, chain0=chain0)
    assert draws.shape == (2, d, C) and np.isfinite(draws).all()
    t = orc.TargetSpec(orc.TARGET_DIAG, d, prec=prec, W=4)
    for c in [0, 1, 77777, C - 1]:
        s = orc.make_settings(seed=8, n_burnin=2, n_keep=2, n_leap=32, step=0.005, W=4, chain_id=chain0 + c)
        o, info = orc.run_chain(orc.ALGO_HMC, t, init[c], s)
        assert np.array_equal(draws[:, :, c], o) and g["n_accept"][c] == info["n_accept"]
    # started in stationarity: the scaled second moment stays 1
    m2 = (draws[-1] ** 2 * prec[:, None]).mean()
    assert abs(m2 - 1) < 0.02


def test_config3_shape_cut_into_pieces_equals_the_tick_local_kernel():
    """65 536 chains on the chip's 16 384 chain slots: the runs are cut into four pieces that migrate between slots -- and XCDs -- through memory inside one launch
    (nuts_memo_core.hpp, SPLIT).  Device-resident, 8 + 8 draws, four chains started non-finite; every output equals the tick-local kernel's (whole chains in fixed
    slots, an independent implementation), and five chains are re-run by the oracle."""
    import torch
    d, C, half = 128, 65536, 8
    dev = torch.device("cuda", 0)
    prec_h = synth.dense_gaussian_precision(d)
    prec = torch.from_numpy(prec_h).to(dev)
    init = synth.initial_states(C, d, seed=3)
    init[5] *= 1e300; init[20000, 7] = np.inf; init[40000, 100] = np.nan; init[65535] *= 1e160
    theta0 = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    st = mcmc_amd.default_settings(rng_seed_value=2024, n_burnin_draws=half, n_keep_draws=half, n_adapt_draws=half)

    def run(hint):
        theta = theta0.clone()
        draws = torch.empty((half, d, C), dtype=torch.float64, device=dev)
        n_leap = torch.zeros(C, dtype=torch.int64, device=dev); eps = torch.zeros(C, dtype=torch.float64, device=dev)
        t = mcmc_amd.make_target(mcmc_amd.TARGET_GAUSS_DENSE, d, prec=prec, mem=mcmc_amd.MEM_DEVICE, kernel_hint=hint)
        ch = mcmc_amd.make_chains(theta, C, draws=draws, n_leapfrogs=n_leap, step_size=eps, mem=mcmc_amd.MEM_DEVICE)
        mcmc_amd.run("nuts", t, st, ch, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return draws, theta, n_leap, eps

    ref = run(mcmc_amd.KERNEL_NUTS_TICK_LOCAL)
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_async_kernel<")
    got = run(mcmc_amd.KERNEL_AUTO)
    assert mcmc_amd.last_kernel().startswith("nuts_gauss_memo_kernel<")
    same = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=123.0, posinf=1e308, neginf=-1e308), torch.nan_to_num(b, nan=123.0, posinf=1e308, neginf=-1e308)))
    assert same(got[0], ref[0]) and same(got[1], ref[1]) and bool(torch.equal(got[2], ref[2])) and same(got[3], ref[3])
    draws = got[0].cpu().numpy()
    t = orc.TargetSpec(orc.TARGET_DENSE, d, prec=prec_h, W=4)
    for c in [0, 16383, 16384, 40001, 65534]:
        s = orc.make_settings(seed=2024, n_burnin=half, n_keep=half, n_adapt=half, step=1.0, W=4, chain_id=c)
        o, info = orc.run_chain(orc.ALGO_NUTS, t, init[c], s, traces=True)
        assert np.array_equal(draws[:, :, c], o) and int(got[2][c].item()) == info["n_leap"] and float(got[3][c].item()) == info["eps"]
