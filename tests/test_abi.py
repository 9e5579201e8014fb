"""CPU tests of the drop-in boundary: libmcmcpp_hip.so loads without a GPU, exports every symbol
include/mcmcpp_hip.h declares, validates arguments like the reference's asserts, and fails loudly
(no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import pytest

from mcmcpp_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return capi.lib()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "mcmcpp_hip.h")).read()
    declared = set(re.findall(r"\b(mcmcpp_hip_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mcmcpp_hip_abi_version() == 2


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/mcmcpp_hip.h compiles as C99 (any FFI generator can read it)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "mcmcpp_hip.h"\nint main(void) { mcmcpp_hip_config c; c.mover = MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION; '
                   'return (int)sizeof(c) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "hdr.o")])


def test_config_struct_matches_header(lib):
    # struct_size is checked by the library: a python/C layout mismatch would be rejected here
    cfg = capi.Config(C.sizeof(capi.Config) - 4, 0, 64, 4, 0, 0, None, 0, 0, -1, 0, 0, 0, 0, 0, None, None, 0, 0)
    h = C.c_void_p()
    assert lib.mcmcpp_hip_create(C.byref(cfg), C.byref(h)) == 1
    assert b"struct_size" in lib.mcmcpp_hip_last_error(None)


@pytest.mark.parametrize("kw,msg", [
    (dict(W=63, D=4), "even"),                      # EnsembleSampler.h:207
    (dict(W=8, D=4), "exceed"),                     # EnsembleSampler.h:208
    (dict(W=64, D=0), "num_params"),
    (dict(W=64, D=4, calc_id=9), "calc_id"),
    (dict(W=64, D=4, calc_id=capi.CALC_DENSE_GAUSSIAN), "D*D"),
    (dict(W=64, D=3, calc_id=capi.CALC_SKEWED_GAUSSIAN_2D, params=[0.13]), "D == 2"),
    (dict(W=64, D=4, dtype=7), "dtype"),
])
def test_argument_errors(lib, kw, msg):
    kw.setdefault("calc_id", capi.CALC_ISO_GAUSSIAN)
    with pytest.raises(capi.HipError) as e:
        capi.HipSampler(kw.pop("W"), kw.pop("D"), kw.pop("calc_id"), **kw)
    assert e.value.code == 1 and msg in str(e.value)


def test_null_handle_is_an_error_not_a_crash(lib):
    assert lib.mcmcpp_hip_run(None, 1, 1, None, None) == 1
    assert lib.mcmcpp_hip_set_state(None, None, None) == 1
    lib.mcmcpp_hip_destroy(None)


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.HipError) as e:
        capi.HipSampler(64, 4, capi.CALC_ISO_GAUSSIAN)
    assert e.value.code in (2, 3)  # HIP error / no device -- never a silent CPU path
