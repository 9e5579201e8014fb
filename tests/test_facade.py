"""The header-only C++ facade (include/MCMCpp): the Chain and its iterators are unit-tested on the CPU, the
samplers are compiled and linked here and run on the GPU against the reference's golden fixtures."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mcmcpp_amd import capi
from tests.goldens import GOLDEN_DIR, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
INC = ["-I" + os.path.join(ROOT, "include", "MCMCpp"), "-I" + os.path.join(ROOT, "include")]
LINK = ["-L" + os.path.join(ROOT, "mcmcpp_amd"), "-lmcmcpp_hip", "-Wl,-rpath," + os.path.join(ROOT, "mcmcpp_amd")]


def _compile(src, out, link):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, out)
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(dp, f))
                                            for dp, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs])
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        capi.build_library()
        cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror"] + INC + [src, "-o", exe] + (LINK if link else [])
        subprocess.check_call(cmd)
    return exe


def test_chain_and_iterators():
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "chain_test.cpp"), "chain_test", link=False)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "chain_test OK" in out.stdout, out.stdout + out.stderr


def test_facade_compiles_and_links_against_the_c_abi():
    _compile(os.path.join(ROOT, "tests", "cpp", "facade_parity.cpp"), "facade_parity", link=True)
    _compile(os.path.join(ROOT, "examples", "skewed_gaussian_stretch.cpp"), "skewed_gaussian_stretch", link=True)
    _compile(os.path.join(ROOT, "examples", "skewed_gaussian_diffevo.cpp"), "skewed_gaussian_diffevo", link=True)


def test_non_device_calculator_is_rejected_at_compile_time(tmp_path):
    src = tmp_path / "bad.cpp"
    src.write_text('#include "Movers/StretchMove.h"\n'
                   'struct HostOnly { double calcLogPostProb(double* p) { return -p[0] * p[0]; } };\n'
                   'int main() { HostOnly c; MCMC::Mover::StretchMove<double, HostOnly> m(1, 0, c); return m.getNumParams(); }\n')
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only"] + INC + [str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "device functor" in r.stderr


def _write_fixture(g, path):
    kept = [k for k in g.full_steps]
    with open(path, "wb") as f:
        npar = 0 if g.params is None else g.params.size
        f.write(struct.pack("8i", g.W, g.D, g.steps, g.slicing, g.calc, npar, len(kept), g.dtype))
        if npar:
            f.write(np.asarray(g.params, dtype=g.np_t).tobytes())
        f.write(np.asarray(g.init_pos, dtype=g.np_t).tobytes())
        f.write(np.asarray(g.init_logp, dtype=g.np_t).tobytes())
        f.write(np.asarray(kept, dtype=np.int32).tobytes())
        for k in kept:
            f.write(np.asarray(g.z["chain_step_%d" % k], dtype=g.np_t).tobytes())
        f.write(struct.pack("2Q", g.accepted_total, g.total_steps))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["iso64x4", "iso100x7", "dense96x16", "rosen80x8", "skewed320x2", "iso64x4_f32",
                                  "dense80x5_f32"])
def test_facade_matches_reference_golden(name, tmp_path):
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "facade_parity.cpp"), "facade_parity", link=True)
    fx = tmp_path / (name + ".bin")
    _write_fixture(Golden(name), fx)
    out = subprocess.run([exe, str(fx)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "facade_parity OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["de_iso64x4", "de_iso100x7", "de_rosen80x8", "de_dense96x16", "de_dense80x5_f32"])
def test_facade_differential_evolution_matches_reference_golden(name, tmp_path):
    """The same program with Mover::DifferentialEvolution (include/MCMCpp/Movers/DifferentialEvolution.h)."""
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "facade_parity.cpp"), "facade_parity", link=True)
    fx = tmp_path / (name + ".bin")
    _write_fixture(Golden(name), fx)
    out = subprocess.run([exe, str(fx), "de"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "facade_parity OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_example_reproduces_reference_test_output(tmp_path):
    """examples/skewed_gaussian_stretch.cpp with the reference test's initial placement prints the
    reference's acceptance line (accepted/total)."""
    import json
    want = json.load(open(os.path.join(GOLDEN_DIR, "reference_skewed_test.json")))
    exe = _compile(os.path.join(ROOT, "examples", "skewed_gaussian_stretch.cpp"), "skewed_gaussian_stretch", link=True)
    init = tmp_path / "init.bin"
    init.write_bytes(np.asarray(Golden("skewed320x2").init_pos, dtype=np.float64).tobytes())
    out = subprocess.run([exe, str(want["stored_steps"]), str(init)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Acceptance Fraction: %d/%d" % (want["accepted_total"], want["total_steps"]) in out.stdout, out.stdout


@pytest.mark.gpu
def test_diffevo_example_reproduces_reference_test_output(tmp_path):
    """examples/skewed_gaussian_diffevo.cpp -- the reference's SkewedGaussian/DiffEvo test (320 x 2, slicing 10, 40 019 stored
    steps) -- with the reference test's initial placement prints the acceptance line the reference computes."""
    import json
    want = json.load(open(os.path.join(GOLDEN_DIR, "reference_skewed_diffevo_test.json")))
    exe = _compile(os.path.join(ROOT, "examples", "skewed_gaussian_diffevo.cpp"), "skewed_gaussian_diffevo", link=True)
    init = tmp_path / "init.bin"
    init.write_bytes(np.asarray(Golden("skewed320x2").init_pos, dtype=np.float64).tobytes())
    out = subprocess.run([exe, str(want["stored_steps"]), str(init)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Acceptance Fraction: %d/%d" % (want["accepted_total"], want["total_steps"]) in out.stdout, out.stdout
