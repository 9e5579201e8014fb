"""mcmcpp_amd -- MI355X (gfx950) implementation of the stretch-move ensemble step of jmatta1/MCMCpp.

The product is the C-ABI shared library ``mcmcpp_amd/libmcmcpp_hip.so`` (sources in ``mcmcpp_amd/csrc``,
interface in ``include/mcmcpp_hip.h``) and the header-only C++ facade in ``include/MCMCpp``.  This
Python package is plumbing for tests and benchmarks: a ctypes binding (``capi``) and the build driver.
"""
from . import capi  # noqa: F401
from .capi import HipSampler, build_library, library_path  # noqa: F401
