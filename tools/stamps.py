#!/usr/bin/env python3
"""Diagnostic: where one wavefront of a half-step launch spends its cycles (needs make -C mcmcpp_amd/csrc STAMPS=1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MCMCPP_HIP_LIB"] = os.path.join(ROOT, "mcmcpp_amd", "libmcmcpp_hip_stamps.so")
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi
from mcmcpp_amd import workloads
names = ["entry->partner gather issued", "tables+next draws", "wait partner rows", "calculator", "accept+stores issued"]
for calc, W in [("iso", 16384), ("dense", 16384), ("iso", 128)]:
    D = 32
    P = workloads.ar1_precision(D, 0.5)
    cid, prm = {"dense": (capi.CALC_DENSE_GAUSSIAN, P.ravel()), "iso": (capi.CALC_ISO_GAUSSIAN, None)}[calc]
    s = capi.HipSampler(W, D, cid, prm, seed=0)
    pos = workloads.init_positions(W, D)
    s.set_state(pos, s.calc_logp(pos))
    acc = []
    for rep in range(20):
        s.run(40, save_chain=False, want_accepted=False)
        out = (C.c_ulonglong * 8)()
        capi.lib().mcmcpp_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
        assert capi.lib().mcmcpp_hip_debug_stamps(s.h, out) == 0
        t = np.array(list(out)[:6], dtype=np.float64)
        acc.append(np.append(np.diff(t), [(out[5] - out[0]) / max(1, (out[7] - out[6])) * 100.0]))
    d = np.median(np.array(acc), axis=0)
    print(calc, W, " | ".join("%s %.0f" % (n, x) for n, x in zip(names, d[:5])), "| total %.0f ticks (s_memtime) | shader clock ~%.0f MHz" % (d[:5].sum(), d[5]))
