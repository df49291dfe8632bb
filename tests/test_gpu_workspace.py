"""GPU: the per-(device, stream) workspace cache -- reuse, release, and two host threads on one stream."""
import threading

import numpy as np
import pytest

import mcmc_amd
from mcmc_amd import synth

pytestmark = pytest.mark.gpu


def test_release_workspace_frees_and_the_next_call_reallocates():
    d, C = 64, 200
    prec = synth.dense_gaussian_precision(d, seed=1)
    init = synth.initial_states(C, d, seed=2)
    st = mcmc_amd.default_settings(rng_seed_value=3, n_burnin_draws=2, n_keep_draws=3, n_adapt_draws=2, max_tree_depth=4)
    a, _ = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)          # the NUTS workspace is the big one
    freed = mcmc_amd.release_workspace()
    assert freed >= 64 * 64 * 8 * (C // 16)                                          # 64 record vectors per chain tile
    assert mcmc_amd.release_workspace() == 0                                         # nothing left
    b, _ = mcmc_amd.nuts(mcmc_amd.TARGET_GAUSS_DENSE, init, st, prec=prec)
    assert np.array_equal(a, b)


def test_two_host_threads_on_the_default_stream_share_the_cache_safely():
    d = 32
    prec = synth.dense_gaussian_precision(d, seed=1)
    st = mcmc_amd.default_settings(rng_seed_value=5, n_burnin_draws=2, n_keep_draws=4, n_leap_steps=3, step_size=0.1)
    want = {C: mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, synth.initial_states(C, d, seed=C), st, prec=prec)[0] for C in (48, 700)}
    mcmc_amd.release_workspace()
    errs = []

    def work(C):
        try:
            for _ in range(6):                                                       # alternating sizes force regrowth under contention
                got, _ = mcmc_amd.hmc(mcmc_amd.TARGET_GAUSS_DENSE, synth.initial_states(C, d, seed=C), st, prec=prec)
                if not np.array_equal(got, want[C]):
                    errs.append(C)
        except Exception as e:                                                      # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(C,)) for C in (48, 700, 48, 700)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
