"""Synthetic workloads of the benchmark: initial walker positions and calculator parameters.

Pure integer-hash / closed-form recipes, exactly reproducible on any IEEE machine; the oracle carries the same
recipe in C (oracle/stretch_oracle.c: so_init_positions) and tests/test_workloads.py checks the two agree.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def init_positions(W, D, salt=0, dtype=np.float64):
    """positions[w, p] = 4*u - 2 with u = top 53 bits of splitmix64(w*D + p + salt*K) * 2^-53: uniform in [-2, 2)."""
    with np.errstate(over="ignore"):
        k = np.arange(W * D, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x632BE59BD9B4E019)
        h = _splitmix64(k)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (4.0 * u - 2.0).astype(dtype).reshape(W, D)


def ar1_precision(D, rho, dtype=np.float64):
    """Dense precision matrix of the AR(1)-correlated Gaussian Sigma_ij = rho^|i-j| (closed-form tridiagonal inverse)."""
    t = np.dtype(dtype).type
    P = np.zeros((D, D), dtype=dtype)
    rho = t(rho)
    d = t(1) - rho * rho
    for i in range(D):
        P[i, i] = (t(1) if i in (0, D - 1) else t(1) + rho * rho) / d
        if i + 1 < D:
            P[i, i + 1] = P[i + 1, i] = -rho / d
    return P
