"""User-compiled device Calculators: a plug-in built with hipcc against mcmcpp_amd/csrc/mcmcpp_hip_plugin.hpp,
registered through mcmcpp_hip_register_calculator and selected by its calculator id."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "plugin_calc.hip")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libplugin_calc.so")
ISO_CLONE, DIAG_SHIFTED = 1000, 1001


@pytest.fixture(scope="module")
def plugin():
    capi.build_library()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr_dir = os.path.join(ROOT, "mcmcpp_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(hdr_dir, f)) for f in os.listdir(hdr_dir) if f.endswith((".hpp", ".inc")))
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(newest, os.path.getmtime(SRC)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-fno-fast-math", "-fPIC", "-shared", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                               "-I" + hdr_dir, SRC, "-o", OUT])
    lib = C.CDLL(OUT)
    tables = {}
    for name in ("iso_clone", "diag_shifted"):
        for t in ("f64", "f32"):
            f = getattr(lib, "mcmcpp_hip_plugin_%s_%s" % (name, t))
            f.restype = C.c_void_p
            tables[name, t] = f()
            assert tables[name, t]
    L = capi.lib()
    L.mcmcpp_hip_register_calculator.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    assert L.mcmcpp_hip_register_calculator(ISO_CLONE, tables["iso_clone", "f64"], tables["iso_clone", "f32"], 0) == 0
    assert L.mcmcpp_hip_register_calculator(DIAG_SHIFTED, tables["diag_shifted", "f64"], tables["diag_shifted", "f32"], -1) == 0
    return L


def test_registration_rules(plugin):
    assert plugin.mcmcpp_hip_register_calculator(5, None, None, 0) == 1          # ids below 1000 are the library's
    assert plugin.mcmcpp_hip_register_calculator(2000, None, None, 0) == 1       # at least one table
    with pytest.raises(capi.HipError) as e:
        capi.HipSampler(64, 4, 1234)                                             # never registered
    assert "unknown calc_id" in str(e.value)
    with pytest.raises(capi.HipError) as e:
        capi.HipSampler(64, 4, ISO_CLONE, params=[1.0])                          # declared to take no parameters
    assert "takes 0 parameters" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("mover", [capi.MOVER_STRETCH, capi.MOVER_DIFFERENTIAL_EVOLUTION])
@pytest.mark.parametrize("dtype", [po.F64, po.F32])
def test_plugin_clone_follows_the_builtin_trajectory(plugin, dtype, mover):
    # (a plug-in's launch table carries the kernels of both movers)
    W, D = 2048, 24
    pos = po.init_positions(dtype, W, D, salt=6)
    builtin = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, seed=2, dtype=dtype, mover=mover)
    clone = capi.HipSampler(W, D, ISO_CLONE, seed=2, dtype=dtype, mover=mover)
    logp = builtin.calc_logp(pos)
    np.testing.assert_array_equal(clone.calc_logp(pos), logp)
    builtin.set_state(pos, logp)
    clone.set_state(pos, logp)
    c1, a1 = builtin.run(40)
    c2, a2 = clone.run(40)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(a1, a2)


@pytest.mark.gpu
def test_plugin_functor_the_library_does_not_ship(plugin):
    W, D = 4096, 10
    rng = np.random.default_rng(0)
    mu = rng.uniform(-3, 3, D)
    w = rng.uniform(0.5, 4.0, D)          # precisions: variance 1/w
    s = capi.HipSampler(W, D, DIAG_SHIFTED, params=np.concatenate([mu, w]), seed=1)
    pos = po.init_positions(po.F64, W, D, salt=1)
    want = -0.5 * (w * (pos - mu) ** 2).sum(axis=1)
    got = s.calc_logp(pos)
    np.testing.assert_allclose(got, want, rtol=1e-13)
    s.set_state(pos, got)
    s.run(1, interval=400, save_chain=False)
    chain, acc = s.run(4, interval=25)
    x = chain.reshape(-1, D)
    assert np.abs(x.mean(axis=0) - mu).max() < 0.05
    assert np.abs(x.var(axis=0) * w - 1).max() < 0.08
    assert 0.3 < acc.mean() / W < 0.6
    assert s.counters()["near_ties"] == 0


def _build_user_calculator_example():
    """examples/user_calculator_device.hip -> libuser_calculator.so (hipcc), examples/user_calculator.cpp -> the program (g++)."""
    capi.build_library()
    build = os.path.dirname(OUT)
    os.makedirs(build, exist_ok=True)
    hdr_dir = os.path.join(ROOT, "mcmcpp_amd", "csrc")
    dev_src = os.path.join(ROOT, "examples", "user_calculator_device.hip")
    host_src = os.path.join(ROOT, "examples", "user_calculator.cpp")
    dev_so = os.path.join(build, "libuser_calculator.so")
    exe = os.path.join(build, "user_calculator")
    newest_hdr = max(os.path.getmtime(os.path.join(hdr_dir, f)) for f in os.listdir(hdr_dir) if f.endswith((".hpp", ".inc")))
    if not os.path.exists(dev_so) or os.path.getmtime(dev_so) < max(newest_hdr, os.path.getmtime(dev_src)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                               "-shared", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-I" + hdr_dir, dev_src, "-o", dev_so])
    newest_inc = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs)
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(newest_inc, os.path.getmtime(host_src), os.path.getmtime(dev_so)):
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include", "MCMCpp"),
                               "-I" + os.path.join(ROOT, "include"), host_src, "-o", exe, "-L" + build, "-luser_calculator",
                               "-L" + os.path.join(ROOT, "mcmcpp_amd"), "-lmcmcpp_hip", "-Wl,-rpath," + build,
                               "-Wl,-rpath," + os.path.join(ROOT, "mcmcpp_amd")])
    return exe


def test_user_calculator_example_builds():
    """CPU: the user's functor compiles against the plug-in header (hipcc cross-compiles) and the C++ program links."""
    _build_user_calculator_example()


@pytest.mark.gpu
def test_user_calculator_through_the_cpp_facade(tmp_path):
    """A host Calculator class of the user's own (hipCalcId = 1000) with its hipcc-built functor, run by
    ParallelEnsembleSampler<double, StretchMove<double, UserClass>> (the reference's template surface,
    /root/reference/MCMCpp/Movers/StretchMove.h:42-54): with means 0 / precisions 1 its chain must be the oracle's chain of
    the isotropic Gaussian it then equals, stored step by stored step; with parameters of its own the program checks the
    sample moments."""
    exe = _build_user_calculator_example()
    W, D, steps = 2048, 24, 60
    pos = po.init_positions(po.F64, W, D, salt=6)
    init = tmp_path / "init.bin"
    init.write_bytes(pos.tobytes())
    out_file = tmp_path / "chain.bin"
    r = subprocess.run([exe, str(W), str(D), str(steps), str(init), str(out_file)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "user_calculator OK" in r.stdout and "identical chains" in r.stdout, r.stdout + r.stderr
    chain = np.fromfile(out_file, dtype=np.float64).reshape(steps + 1, W, D)
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=0)
    orc.set_state(pos, orc.logp(pos))
    want, _ = orc.run(steps, mode=po.MODE_COUNTER, threads=4)
    np.testing.assert_array_equal(chain[0], pos)
    np.testing.assert_array_equal(chain[1:], want)
