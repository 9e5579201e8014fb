"""Chain analysis: Analysis::CovarianceMatrix.

CPU: the oracle's restatement (oracle/stretch_oracle_typed.inc: chain_covariance) against fixtures produced by the
reference's own CovarianceMatrix over its own Chain (tests/golden/make_golden.py covariance) -- bit for bit.
GPU: the device accumulation (mcmcpp_hip_moments_*, matrix cores) against the oracle.  The reference's sequential
Kahan sums cannot be kept by a parallel sum, so the bar is a stated tolerance: every covariance element within
1e-10 * sqrt(var_i var_j) (fp64) of the oracle's, every correlation element within 1e-10; for fp32 chains 2e-4,
which is the accuracy of the reference's own fp32 arithmetic (the device accumulates fp32 samples in fp64 and is the
more accurate of the two: it is also checked against an fp64 computation at 1e-6)."""
import os
import subprocess

import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po
from tests.goldens import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = ["covariance_dense96x16", "covariance_dense96x16_slice5", "covariance_rosen80x8", "covariance_dense80x5_f32"]


def _fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z["steps"], int(z["slice_interval"]), z["cov"], z["corr"]


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_covariance_matches_the_reference_bit_for_bit(name):
    steps, sl, cov, corr = _fixture(name)
    mean, c, r = po.chain_covariance(steps, sl)
    np.testing.assert_array_equal(c, cov)
    np.testing.assert_array_equal(r, corr)
    used = steps[::sl].reshape(-1, steps.shape[2]).astype(np.float64)
    np.testing.assert_allclose(mean, used.mean(axis=0), rtol=1e-5 if steps.dtype == np.float32 else 1e-12, atol=1e-6)


def _close(got_cov, got_corr, cov, corr, tol):
    scale = np.sqrt(np.abs(np.outer(np.diag(cov), np.diag(cov)))).astype(np.float64)
    assert np.all(np.abs(got_cov.astype(np.float64) - cov) <= tol * scale), np.abs(got_cov - cov).max()
    assert np.all(np.abs(got_corr.astype(np.float64) - corr) <= tol), np.abs(got_corr - corr).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_device_covariance_matches_the_reference_fixture(name):
    steps, sl, cov, corr = _fixture(name)
    f32 = steps.dtype == np.float32
    m = capi.HipMoments(steps.shape[1], steps.shape[2], dtype=capi.F32 if f32 else capi.F64)
    m.add_steps(steps, sl)
    n, mean, c, r = m.finish()
    assert n == len(steps[::sl]) * steps.shape[1]
    _close(c, r, cov, corr, 2e-4 if f32 else 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("W,D,n,sl,dt", [
    (10, 1, 50, 1, np.float64), (34, 2, 40, 3, np.float64), (100, 7, 30, 1, np.float64), (64, 16, 25, 2, np.float64),
    (70, 17, 20, 1, np.float64), (4098, 32, 6, 1, np.float64), (130, 33, 12, 5, np.float64), (200, 64, 9, 2, np.float64),
    (150, 65, 7, 1, np.float64),      # beyond the matrix-core path: generic kernel
    (300, 130, 4, 1, np.float64), (96, 5, 60, 2, np.float32), (128, 32, 20, 1, np.float32), (40, 70, 10, 3, np.float32)])
def test_device_covariance_matches_the_oracle(W, D, n, sl, dt):
    rng = np.random.default_rng(W * 131 + D)
    mix = rng.standard_normal((D, D)) / np.sqrt(D) + np.eye(D)
    steps = (rng.standard_normal((n, W, D)) @ mix + rng.standard_normal(D) * 3).astype(dt)
    f32 = dt == np.float32
    _, cov, corr = po.chain_covariance(steps, sl)
    m = capi.HipMoments(W, D, dtype=capi.F32 if f32 else capi.F64)
    m.add_steps(steps, sl)
    npts, mean, c, r = m.finish()
    assert npts == len(steps[::sl]) * W
    _close(c, r, cov, corr, 2e-4 if f32 else 1e-10)
    # against an independent fp64 computation (for fp32 chains the device is the more accurate of the two)
    used = steps[::sl].reshape(-1, D).astype(np.float64)
    ref = np.cov(used.T, bias=True).reshape(D, D)
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert np.all(np.abs(c - ref) <= (1e-6 if f32 else 1e-10) * scale + (1e-6 if f32 else 0) * np.abs(np.outer(used.mean(0), used.mean(0))))
    np.testing.assert_allclose(mean, used.mean(axis=0), rtol=1e-6 if f32 else 1e-12, atol=1e-6 if f32 else 1e-13)


@pytest.mark.gpu
def test_device_covariance_accumulates_over_calls_and_resets():
    rng = np.random.default_rng(5)
    steps = rng.standard_normal((40, 256, 32)) * np.linspace(0.5, 2.0, 32)
    _, cov, corr = po.chain_covariance(steps, 1)
    m = capi.HipMoments(256, 32)
    for part in (steps[:7], steps[7:8], steps[8:29], steps[29:]):      # chain blocks arriving one by one
        m.add_steps(part)
    _, _, c, r = m.finish()
    _close(c, r, cov, corr, 1e-10)
    again = m.finish()                                                   # finish does not consume the sums
    np.testing.assert_array_equal(again[2], c)
    m.reset()
    m.add_steps(steps[:10])
    _, cov10, corr10 = po.chain_covariance(steps[:10], 1)
    _, _, c10, r10 = m.finish()
    _close(c10, r10, cov10, corr10, 1e-10)
    with pytest.raises(capi.HipError):
        capi.HipMoments(256, 32).finish()                                # nothing added


@pytest.mark.gpu
def test_device_resident_steps_give_the_same_sums():
    import torch
    rng = np.random.default_rng(9)
    steps = rng.standard_normal((6, 1024, 32)) + 0.5
    a, b = capi.HipMoments(1024, 32), capi.HipMoments(1024, 32)
    a.add_steps(steps)
    t = torch.from_numpy(steps).cuda()
    b.add_device_steps(t.data_ptr(), steps.shape[0])
    for x, y in zip(a.finish(), b.finish()):
        np.testing.assert_array_equal(x, y)                               # same kernel, same order: bit-identical


@pytest.mark.gpu
def test_covariance_facade_against_the_oracle():
    """include/MCMCpp/Analysis/CovarianceMatrix.h on a chain sampled through the facade (tests/cpp/covariance_facade.cpp)."""
    from tests.test_facade import BUILD, INC, LINK
    po.build()
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "covariance_facade")
    src = os.path.join(ROOT, "tests", "cpp", "covariance_facade.cpp")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(dp, f))
                                            for dp, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs])
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        capi.build_library()
        oracle_dir = os.path.join(ROOT, "oracle")
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror"] + INC + [src, "-o", exe] + LINK +
                              ["-L" + oracle_dir, "-loracle", "-Wl,-rpath," + oracle_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "covariance_facade OK" in out.stdout, out.stdout + out.stderr
