"""Effective sample size of many-chain draws (the reference has no ESS code; SURVEY.md 8(d) defines it here).

Per dimension: autocovariances are averaged over chains (chains are i.i.d. replicas of one process), the
autocorrelation sum is truncated by Geyer's initial positive sequence, ESS_chain = n / (1 + 2 sum rho_t),
and the many-chain ESS is C * ESS_chain.  Reported: the minimum over dimensions."""
import numpy as np


def ess_per_chain(draws):
    """draws: [n_keep, d, C] -> array [d] of per-chain ESS (pooled autocovariance over the C chains)."""
    x = np.asarray(draws, dtype=np.float64)
    n, d, C = x.shape
    if n < 4:
        return np.full(d, float(n))
    x = x - x.mean(axis=(0, 2), keepdims=True)
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(x, n=nfft, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:n].mean(axis=2)        # [n, d], summed lags
    acov /= np.arange(n, 0, -1)[:, None]                                        # unbiased per lag
    rho = acov / np.where(acov[0] > 0, acov[0], 1.0)
    out = np.empty(d)
    for j in range(d):
        r = rho[:, j]
        tau = -1.0
        t = 0
        while t + 1 < n:                       # Geyer: sums of adjacent pairs stay positive
            pair = r[t] + r[t + 1]
            if pair <= 0:
                break
            tau += 2.0 * pair
            t += 2
        out[j] = n / max(tau, 1.0 / n) if tau > 0 else float(n)
    return np.minimum(out, n * 10.0)


def ess_min_total(draws, n_chains_total=None):
    """min over dims of the per-chain ESS, times the number of chains it stands for."""
    e = ess_per_chain(draws)
    C = draws.shape[2] if n_chains_total is None else n_chains_total
    return float(e.min()) * C, e
