"""The split ensemble with MORE THAN ONE RANK, executed on the one GPU a test box has: G handles on G host threads
exchange through a loop-back collective library (tests/cpp/loopback_ccl.hip) that libmcmcpp_hip.so loads in place of
RCCL (MCMCPP_HIP_RCCL_LIB).  This runs the product's world > 1 code -- run_split, the exchanges' offsets and their order
relative to the launches, the status agreement, the facade's one-thread-per-rank constructor
(reference: ParallelEnsembleSampler.h:228-262,285-291; Threading/RedBlkCtrlerSpinLock.h:240-322) -- bit for bit against
the oracle.  It validates the exchange logic; RCCL's own transport between GPUs is not exercised here.

The collective library is bound once per process, so the cases run in ONE child process (tests/loopback_driver.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mcmcpp_amd import capi
from tests.goldens import Golden
from tests.test_facade import _compile, _write_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
SHIM_SRC = os.path.join(ROOT, "tests", "cpp", "loopback_ccl.hip")
SHIM = os.path.join(BUILD, "libloopback_ccl.so")

sys.path.insert(0, os.path.join(ROOT, "tests"))
import loopback_driver  # noqa: E402  (the list of cases; nothing runs on import)


def build_shim():
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < os.path.getmtime(SHIM_SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-Wall",
                               SHIM_SRC, "-o", SHIM])
    return SHIM


def test_loopback_library_exports_what_the_product_binds():
    """CPU: the shim builds (hipcc cross-compiles) and carries every nccl* symbol rccl_dyn.hpp looks up."""
    shim = build_shim()
    names = subprocess.run(["nm", "-D", "--defined-only", shim], capture_output=True, text=True, check=True).stdout
    src = open(os.path.join(ROOT, "mcmcpp_amd", "csrc", "rccl_dyn.hpp")).read()
    import re
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(wanted) >= 10
    for w in wanted:
        assert " T %s\n" % w in names, w


_results = {}


def _run_driver():
    if _results:
        return _results
    env = dict(os.environ, MCMCPP_HIP_RCCL_LIB=build_shim(), LOOPBACK_CCL_TIMEOUT_S="180")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "loopback_driver.py")], capture_output=True, text=True, timeout=3000, env=env)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            rec = json.loads(line)
            _results[rec["case"]] = rec
    _results["__tail__"] = {"ok": True, "problems": [out.stdout[-1500:], out.stderr[-3000:]], "returncode": out.returncode}
    return _results


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(loopback_driver.CASES))
def test_split_ensemble_over_several_ranks_matches_the_oracle(name):
    res = _run_driver()
    assert name in res, "the driver did not reach this case: %r" % (res["__tail__"],)
    assert res[name]["ok"], res[name]["problems"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,ranks", [("iso64x4", 2), ("dense96x16", 4), ("rosen80x8", 2), ("iso64x4_f32", 2)])
def test_facade_over_several_ranks_matches_reference_golden(name, ranks, tmp_path):
    """The C++ facade with the reference's own user code, the ensemble split over `ranks` ranks on device 0: through
    MCMCPP_DEVICES=0,0,... (unchanged user code: every sampler of the program) and through an explicit Device::Placement."""
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "facade_parity.cpp"), "facade_parity", link=True)
    fx = tmp_path / (name + ".bin")
    _write_fixture(Golden(name), fx)
    env = dict(os.environ, MCMCPP_HIP_RCCL_LIB=build_shim(), LOOPBACK_CCL_TIMEOUT_S="180", MCMCPP_DEVICES=",".join(["0"] * ranks),
               FACADE_PARITY_PLACEMENT_RANKS=str(ranks))
    out = subprocess.run([exe, str(fx)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "facade_parity OK" in out.stdout, out.stdout + out.stderr
    assert "placement of %d ranks checked" % ranks in out.stdout, out.stdout
