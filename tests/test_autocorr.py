"""Chain analysis: Analysis::AutoCorrCalc and Analysis::Detail::AutoCov.

CPU: the oracle's restatement (oracle/stretch_oracle_typed.inc: norm_autocov, autocorr_times) against fixtures the
reference produced (tests/golden/make_golden.py autocorr) -- bit for bit: Detail::AutoCov::calcNormAutoCov on single
series, and the whole AutoCorrCalc class with the oracle's defect emulation on (the reference adds every series onto
the previous walker's result, AutoCorrCalc.h:239-245; see the .inc).
GPU: mcmcpp_hip_autocorr_times against the oracle with the emulation off -- times AND averaged autocovariance
functions bit for bit, both element types, transform in LDS and in global memory, walker subsets, scattered steps."""
import os

import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po
from tests.goldens import GOLDEN_DIR, ar_chain, sha

SERIES = ["f64_n100", "f64_n1000", "f64_n1024", "f64_n1025", "f64_n3", "f32_n777", "f32_n2048"]


@pytest.mark.parametrize("tag", SERIES)
def test_oracle_autocov_matches_the_reference_bit_for_bit(tag):
    z = np.load(os.path.join(GOLDEN_DIR, "autocov_series.npz"))
    dtype = po.F32 if tag.startswith("f32") else po.F64
    got = po.norm_autocov(z[tag + "_series"], float(z[tag + "_avg"]), dtype)
    np.testing.assert_array_equal(got, z[tag + "_autocov"])
    assert got[0] == 1


@pytest.mark.parametrize("name", ["autocorr_class_f64", "autocorr_class_f32"])
def test_oracle_autocorr_class_matches_the_reference_bit_for_bit(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    dtype = int(z["dtype"])
    steps = ar_chain(int(z["n"]), int(z["W"]), int(z["D"]), int(z["seed"]), z["phi"], po.np_dtype(dtype))
    assert sha(steps) == str(z["steps_sha256"])
    emulated = po.autocorr_times(steps, int(z["window_scaling"]), True, dtype)
    np.testing.assert_array_equal(emulated, z["times"])
    # what the class is documented to compute differs from that only through the defect: a fraction of a percent here
    clean = po.autocorr_times(steps, int(z["window_scaling"]), False, dtype)
    np.testing.assert_allclose(clean, z["times"], rtol=5e-3)
    # and both sit at the AR(1) value (1 + phi) / (1 - phi) within the estimator's scatter
    phi = z["phi"]
    np.testing.assert_allclose(clean, (1 + phi) / (1 - phi), rtol=0.12)


@pytest.mark.skipif(not po.reference_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("dtype", [po.F64, po.F32])
def test_oracle_autocov_against_the_live_reference(dtype):
    for n, seed in ((2, 1), (5, 2), (64, 3), (333, 4), (4097, 5)):
        x = ar_chain(n, 1, 1, seed, 0.8, po.np_dtype(dtype)).ravel()
        avg = float(po.np_dtype(dtype)(x.astype(np.float64).mean()))
        np.testing.assert_array_equal(po.norm_autocov(x, avg, dtype), po.reference_norm_autocov(x, avg, dtype))


ANALYTIC = np.array([9.0, 20.0, 30.0, 60.0, 200.0])  # (1 + phi) / (1 - phi) of the test's five phi, as its source states


@pytest.mark.parametrize("name", ["actime_65535", "actime_262143"])
def test_oracle_reproduces_the_references_own_autocorr_test(name):
    """test/sequential/AcTime/src/main.cpp, the reference's known-answer test of AutoCorrCalc, as the reference runs it
    here (fixture): the oracle regenerates the chain (AutoRegressiveMove + libstdc++'s normal_distribution over pcg64,
    digest pinned) and, emulating transferWalker's accumulation, reproduces the reference's five times bit for bit.
    The numbers recorded in the test's source came from another build; they and the analytic values are met within
    the estimator's scatter."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    chain = po.ar1_test_chain(int(z["n_steps"]), int(z["W"]), tuple(z["phis"]))
    assert sha(chain) == str(z["chain_sha256"])
    np.testing.assert_array_equal(po.autocorr_times(chain, 4, True), z["times"])
    if name == "actime_262143":
        np.testing.assert_allclose(z["times"], z["recorded_in_the_test_source"], rtol=0.015)
        np.testing.assert_allclose(z["times"], ANALYTIC, rtol=0.03)


def test_oracle_window_that_never_closes_returns_the_negated_sum():
    steps = ar_chain(64, 2, 1, 9, 0.99)  # 64 samples of a process with time ~200
    t = po.autocorr_times(steps, 1000)
    assert t[0] < 0


CASES = [  # n, W, D, phi, window, dtype
    (2, 2, 1, 0.5, 4, np.float64), (3, 4, 2, 0.5, 4, np.float64), (100, 6, 3, (0.9, 0.5, 0.1), 4, np.float64),
    (1000, 8, 2, (0.9, 0.6), 5, np.float64), (1024, 4, 5, 0.7, 4, np.float64), (4096, 3, 2, (0.95, 0.2), 4, np.float64),
    (5000, 3, 2, (0.95, 0.2), 4, np.float64),  # fft 8192 doubles: global-memory transform
    (777, 10, 4, 0.8, 4, np.float32), (8192, 2, 2, (0.9, 0.4), 4, np.float32), (9000, 2, 3, 0.9, 3, np.float32),  # the last: global
    (300, 34, 33, 0.6, 4, np.float64),
]


@pytest.mark.gpu
@pytest.mark.parametrize("n,W,D,phi,window,dt", CASES)
def test_device_autocorr_matches_the_oracle_bit_for_bit(n, W, D, phi, window, dt):
    dtype = po.F32 if dt == np.float32 else po.F64
    steps = ar_chain(n, W, D, 100 + n, phi, dt)
    want_t, want_f = po.autocorr_times(steps, window, False, dtype, want_functions=True)
    got_t, got_f = capi.autocorr_times(steps, 0, window, want_functions=True)
    np.testing.assert_array_equal(got_f, want_f)
    np.testing.assert_array_equal(got_t, want_t)


@pytest.mark.gpu
def test_device_passes_the_references_own_autocorr_test():
    """The reference's AcTime test (100 walkers x 5 AR(1) parameters x 262144 stored steps, 1 GB) on the device: bit
    for bit the oracle's times (transforms of 2^18 points, in global memory), and the known answers of the test."""
    z = np.load(os.path.join(GOLDEN_DIR, "actime_262143.npz"))
    chain = po.ar1_test_chain(int(z["n_steps"]), int(z["W"]), tuple(z["phis"]))
    assert sha(chain) == str(z["chain_sha256"])
    got = capi.autocorr_times(chain, 0, 4)
    np.testing.assert_array_equal(got, po.autocorr_times(chain, 4, False))
    np.testing.assert_allclose(got, z["recorded_in_the_test_source"], rtol=0.01)  # 9.01951, 19.9437, 29.7831, 59.8488, 196.85
    np.testing.assert_allclose(got, ANALYTIC, rtol=0.02)
    np.testing.assert_allclose(got, z["times"], rtol=0.015)                       # the reference's run, defect and all


@pytest.mark.gpu
def test_device_autocorr_window_that_never_closes():
    steps = ar_chain(64, 2, 2, 9, 0.99)
    got = capi.autocorr_times(steps, 0, 1000)
    np.testing.assert_array_equal(got, po.autocorr_times(steps, 1000))
    assert np.all(got < 0)


@pytest.mark.gpu
@pytest.mark.parametrize("use", [1, 5, 16, 31])
def test_device_autocorr_walker_subset_is_the_evenly_spaced_one(use):
    W, D, n = 32, 3, 500
    steps = ar_chain(n, W, D, 77, (0.9, 0.5, 0.7))
    idx = [(i * W) // use for i in range(use)]
    want = po.autocorr_times(steps[:, idx, :], 4)
    np.testing.assert_array_equal(capi.autocorr_times(steps, use, 4), want)


@pytest.mark.gpu
def test_device_autocorr_steps_scattered_in_host_memory():
    n, W, D = 200, 6, 2
    steps = ar_chain(n, W, D, 5, (0.8, 0.3))
    # blocks of 7 steps with gaps between them, like a chain that lives in several ChainBlocks
    blocks, keep = [], []
    for b in range(0, n, 7):
        buf = np.full((9, W, D), np.nan)
        m = min(7, n - b)
        buf[1:1 + m] = steps[b:b + m]
        keep.append(buf)
        blocks += [buf[1 + i] for i in range(m)]
    np.testing.assert_array_equal(capi.autocorr_times(blocks, 0, 4), po.autocorr_times(steps, 4))


@pytest.mark.gpu
def test_device_autocorr_many_walkers_in_several_passes():
    # 2100 walkers x 16 parameters x 4000 samples: more than one pass over the 1 GiB of functions
    n, W, D = 4000, 2100, 16
    steps = ar_chain(n, W, D, 3, np.linspace(0.2, 0.9, D))
    want = po.autocorr_times(steps, 4)
    np.testing.assert_array_equal(capi.autocorr_times(steps, 0, 4), want)


@pytest.mark.gpu
def test_device_autocorr_on_a_sampler_chain():
    W, D, n = 64, 4, 600
    pos = po.init_positions(po.F64, W, D, salt=2)
    logp = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None).logp(pos)
    s = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=5)
    s.set_state(pos, logp)
    chain, _ = s.run(n, 1)
    got = capi.autocorr_times(chain, 0, 4)
    np.testing.assert_array_equal(got, po.autocorr_times(chain, 4))
    assert np.all(got > 1) and np.all(got < 60)  # the stretch move on an isotropic Gaussian in 4 dimensions


@pytest.mark.gpu
def test_device_resident_chain_gives_the_same_times():
    import torch
    steps = ar_chain(700, 12, 5, 8, (0.9, 0.5, 0.7, 0.2, 0.8))
    t = torch.from_numpy(steps).cuda()
    got = capi.autocorr_times_device(t.data_ptr(), 700, 12, 5)
    np.testing.assert_array_equal(got, capi.autocorr_times(steps, 0, 4))
    np.testing.assert_array_equal(got, po.autocorr_times(steps, 4))


@pytest.mark.gpu
def test_device_autocorr_rejects_bad_arguments():
    steps = ar_chain(10, 2, 2, 1, 0.5)
    with pytest.raises(capi.HipError):
        capi.autocorr_times(steps[:1], 0, 4)
    with pytest.raises(capi.HipError):
        capi.autocorr_times(steps, 3, 4)


def _build_facade_test():
    import subprocess
    from tests.test_facade import BUILD, INC, LINK
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    po.build()
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "autocorr_facade")
    src = os.path.join(root, "tests", "cpp", "autocorr_facade.cpp")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(dp, f))
                                            for dp, _, fs in os.walk(os.path.join(root, "include")) for f in fs])
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        capi.build_library()
        oracle_dir = os.path.join(root, "oracle")
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror"] + INC + [src, "-o", exe] + LINK +
                              ["-L" + oracle_dir, "-loracle", "-Wl,-rpath," + oracle_dir])
    return exe


def test_autocorr_facade_compiles_with_the_reference_signatures():
    assert os.path.exists(_build_facade_test())


@pytest.mark.gpu
def test_autocorr_facade_against_the_oracle():
    """include/MCMCpp/Analysis/AutoCorrCalc.h on a chain sampled through the facade (tests/cpp/autocorr_facade.cpp)."""
    import subprocess
    out = subprocess.run([_build_facade_test()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "autocorr_facade OK" in out.stdout, out.stdout + out.stderr
