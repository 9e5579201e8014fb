"""The benchmark's synthetic inputs (mcmcpp_amd/workloads.py) equal the oracle's C recipe bit for bit."""
import numpy as np
import pytest

from mcmcpp_amd import workloads
from oracle import pyoracle as po


@pytest.mark.parametrize("W,D,salt,dtype", [(64, 4, 0, po.F64), (16384, 32, 0, po.F64), (100, 7, 5, po.F64), (96, 16, 3, po.F32)])
def test_init_positions_match_the_oracle_recipe(W, D, salt, dtype):
    a = workloads.init_positions(W, D, salt, po.np_dtype(dtype))
    b = po.init_positions(dtype, W, D, salt)
    np.testing.assert_array_equal(a, b)


def test_ar1_precision_inverts_the_correlation_matrix():
    D, rho = 32, 0.5
    P = workloads.ar1_precision(D, rho)
    sigma = rho ** np.abs(np.subtract.outer(np.arange(D), np.arange(D)))
    np.testing.assert_allclose(P @ sigma, np.eye(D), atol=1e-12)
