"""GPU: settings.vals_bound on the LDS-streamed kernels (mcmc_amd/csrc/lds_box.hpp; ref: src/hmc.cpp:84-95,107-122,134-136,211-218 and the
same lines of src/nuts.cpp) -- hmc and nuts on the logistic-regression target (8 < d <= 512) and on dense Gaussians with 128 < d <= 512, in
the transformed space with the vectors of a chain split over four waves and log_jacobian's running sum relayed through them.  Identity or
diagonal precond_mat.  Bit for bit against the oracle; the literal kernel (which served these cases before) as a second witness."""
import numpy as np
import pytest

import mcmc_amd
import orc
from mcmc_amd import synth
from test_gpu_parity_nuts_lds import _bs, _problem

pytestmark = pytest.mark.gpu
ALGO = {"hmc": orc.ALGO_HMC, "nuts": orc.ALGO_NUTS}


def _bounds(d, seed, frac=0.3):
    rng = np.random.default_rng(seed)
    kind = np.where(rng.random(d) < frac, rng.integers(2, 5, d), 1)
    kind[0] = 4; kind[d - 1] = 2                                   # the first and the last dimension: the relay starts and ends on a bounded one
    lb = np.where((kind == 2) | (kind == 4), -1.5, -np.inf); ub = np.where((kind == 3) | (kind == 4), 2.0, np.inf)
    return lb, ub


def _check(algo, g_draws, g, o_draws, o):
    assert np.array_equal(g["n_accept"], o["n_accept"])
    assert np.array_equal(g["n_leap"], o["n_leap"])
    assert np.array_equal(g_draws, o_draws, equal_nan=True)
    if algo == "nuts":
        assert np.array_equal(g["eps"], o["eps"], equal_nan=True)


CASES = [("logistic", 20, 37, 37), ("logistic", 100, 16, 5), ("logistic", 200, 30, 37), ("logistic", 512, 24, 5),
         ("dense", 160, 0, 37), ("dense", 256, 0, 5), ("dense", 300, 0, 5), ("dense", 512, 0, 37)]


@pytest.mark.parametrize("algo", ["hmc", "nuts"])
@pytest.mark.parametrize("kind,d,n_rows,C", CASES)
@pytest.mark.parametrize("mass", [False, True])
def test_bounded_hmc_and_nuts_on_the_lds_kernels_match_the_oracle(algo, kind, d, n_rows, C, mass):
    if mass and (d in (20, 200, 256, 300)):
        pytest.skip("the diagonal-mass variant is the same instantiation: half of the shapes carry it")
    tk, tkw, spec = _problem(kind, d, n_rows, seed=d + 2)
    lb, ub = _bounds(d, d)
    init = np.clip(synth.initial_states(C, d, seed=d + 4) * (0.1 if kind == "logistic" else 0.5), -1.0, 1.5)
    kw, okw = dict(vals_bound=1, lower_bounds=lb, upper_bounds=ub), dict(lower=lb, upper=ub)
    if mass:
        M = np.diag(np.random.default_rng(d + 1).uniform(0.4, 2.5, d))
        kw["precond_mat"] = M; okw["precond"] = M
    eps = 0.05 if kind == "logistic" else 0.02
    st = mcmc_amd.default_settings(rng_seed_value=17, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=4, step_size=eps, n_adapt_draws=3,
                                   max_tree_depth=5, **kw)
    g_draws, g = mcmc_amd.sample(algo, tk, init, st, chain0=6, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<") and mcmc_amd.last_kernel().endswith("true, true>")
    s = orc.make_settings(seed=17, n_burnin=2, n_keep=4, n_leap=4, step=eps, n_adapt=3, max_depth=5, W=4, hoist=1, blocks=4,
                          block_size=_bs(kind, d), **okw)
    o_draws, o = orc.run_many(ALGO[algo], spec, init, s, chain0=6)
    assert o["n_accept"].sum() > 0
    assert np.all(g_draws[:, (lb > -np.inf), :] > -1.5 - 1e-12) and np.all(g_draws[:, (ub < np.inf), :] < 2.0 + 1e-12)
    _check(algo, g_draws, g, o_draws, o)


@pytest.mark.parametrize("algo", ["hmc", "nuts"])
@pytest.mark.parametrize("kind,d,n_rows", [("logistic", 300, 40), ("dense", 192, 0)])
def test_bounded_lds_kernels_match_the_literal_kernel_on_longer_runs_and_in_the_non_finite_regime(algo, kind, d, n_rows):
    C = 70
    tk, tkw, _ = _problem(kind, d, n_rows, seed=d + 7)
    lb, ub = _bounds(d, d + 1, frac=0.5)
    init = np.clip(synth.initial_states(C, d, seed=d + 5) * (0.1 if kind == "logistic" else 0.5), -1.0, 1.5)
    init[3, 1:d - 1] *= 1e200                               # unbounded or not: these chains overflow at once (flagged, replayed literally)
    init[40, 0] = 2.0                                       # ON the upper bound of a both-sided dimension: transform = log(eps) - ... finite, huge
    st = mcmc_amd.default_settings(rng_seed_value=23, n_burnin_draws=6, n_keep_draws=6, n_leap_steps=7, step_size=0.05, n_adapt_draws=5,
                                   max_tree_depth=7, vals_bound=1, lower_bounds=lb, upper_bounds=ub)
    a_draws, a = mcmc_amd.sample(algo, tk, init, st, chain0=2, **tkw)
    assert mcmc_amd.last_kernel().startswith("logit_lds_kernel<")
    b_draws, b = mcmc_amd.sample(algo, tk, init, st, chain0=2, kernel_hint=mcmc_amd.KERNEL_LITERAL, **tkw)
    assert mcmc_amd.last_kernel().startswith("literal_kernel<")
    _check(algo, a_draws, a, b_draws, b)
    assert np.array_equal(a["theta"], b["theta"], equal_nan=True)
    # ... and against the ORACLE on the same run (the comparison above is the engine against itself)
    _, _, spec = _problem(kind, d, n_rows, seed=d + 7)
    s = orc.make_settings(seed=23, n_burnin=6, n_keep=6, n_leap=7, step=0.05, n_adapt=5, max_depth=7, W=4, hoist=1, blocks=4,
                          block_size=_bs(kind, d), lower=lb, upper=ub)
    o_draws, o = orc.run_many(ALGO[algo], spec, init, s, chain0=2)
    _check(algo, a_draws, a, o_draws, o)
