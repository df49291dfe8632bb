"""Synthetic inputs of the BASELINE.json configs; numpy only."""
import numpy as np


def dense_gaussian_precision(d, seed=2):
    """C2 / C4 target: P = A A^T / d + I, A_ij ~ N(0,1)."""
    A = np.random.default_rng(seed).standard_normal((d, d))
    P = A @ A.T / d + np.eye(d)
    return np.ascontiguousarray(0.5 * (P + P.T))


def ill_conditioned_diag(d, cond=1.0e4):
    """C5 target: diagonal precision log-spaced in [1, cond]."""
    return np.logspace(0.0, np.log10(cond), d)


def logistic_problem(d, n_rows, seed=4):
    """C3 target: X_ij ~ N(0,1)/sqrt(d), beta* ~ N(0,I), y ~ Bernoulli(sigmoid(X beta*))."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_rows, d)) / np.sqrt(d)
    beta = np.random.default_rng(seed + 1).standard_normal(d)
    p = 1.0 / (1.0 + np.exp(-(X @ beta)))
    y = (np.random.default_rng(seed + 2).random(n_rows) < p).astype(np.float64)
    return np.ascontiguousarray(X), y


def initial_states(n_chains, d, seed=3, chain0=0):
    """theta_0 ~ N(0, I) per chain, [C, d]; row c depends only on the global chain id."""
    out = np.empty((n_chains, d))
    # block-seeded so that a shard [chain0, chain0+C) sees the same rows as the full run
    blk = 4096
    first = (chain0 // blk) * blk
    pos = 0
    b = first
    while pos < n_chains:
        rows = np.random.default_rng([seed, b // blk]).standard_normal((blk, d))
        lo = max(chain0, b) - b
        hi = min(chain0 + n_chains, b + blk) - b
        out[pos:pos + hi - lo] = rows[lo:hi]
        pos += hi - lo
        b += blk
    return out
