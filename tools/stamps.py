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
if os.environ.get("MCMCPP_HIP_FULL_STEP", "1") != "0":  # full-step kernels (full_step_kernel.hpp)
    names = ["entry->2nd trip issued", "->tile 1 done (x|red)", "red decide+stores", "black tile", "black decide+stores"]
# python tools/stamps.py [calc W D f64|f32] ...   (default: the three C2-sized cases)
cases = [("iso", 16384, 32, "f64"), ("dense", 16384, 32, "f64"), ("iso", 128, 32, "f64")]
if len(sys.argv) > 1:
    cases = [(sys.argv[i], int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[i + 3]) for i in range(1, len(sys.argv), 4)]
for calc, W, D, dtype in cases:
    P = workloads.ar1_precision(D, 0.5)
    cid, prm = {"dense": (capi.CALC_DENSE_GAUSSIAN, P.ravel()), "iso": (capi.CALC_ISO_GAUSSIAN, None)}[calc]
    s = capi.HipSampler(W, D, cid, prm, seed=0, dtype=capi.F32 if dtype == "f32" else capi.F64)
    pos = workloads.init_positions(W, D)
    s.set_state(pos, s.calc_logp(pos))
    acc = []
    for rep in range(20):
        s.run(40, save_chain=False, want_accepted=False)
        out = (C.c_ulonglong * (8 + 6 * 4096 + 8))()
        capi.lib().mcmcpp_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
        assert capi.lib().mcmcpp_hip_debug_stamps(s.h, out) == 0
        t = np.array(list(out)[:6], dtype=np.float64)
        acc.append(np.append(np.diff(t), [(out[5] - out[0]) / max(1, (out[7] - out[6])) * 100.0]))
    raw = np.array(list(out)[8:8 + 6 * 4096], dtype=np.int64).reshape(2, 3 * 4096)
    dst = np.array(list(out)[8 + 6 * 4096:8 + 6 * 4096 + 8], dtype=np.int64)
    sets = []
    for k in range(2):
        bk = raw[k, :8192].reshape(-1, 2)
        nb = int((bk[:, 0] > 0).sum())
        sets.append((bk[:nb], raw[k, 8192:8192 + nb]))
    sets.sort(key=lambda x: x[0][:, 0].min() if len(x[0]) else 0)   # earlier launch first
    prev, (blk, dw) = sets[0], sets[1]
    d = np.median(np.array(acc), axis=0)
    print(calc, W, D, dtype, " | ".join("%s %.0f" % (n, x) for n, x in zip(names, d[:5])), "| total %.0f ticks (s_memtime) | shader clock ~%.0f MHz" % (d[:5].sum(), d[5]))
    if len(blk):
        t0 = blk[:, 0].min()
        st, en = (blk[:, 0] - t0) * 10, (blk[:, 1] - t0) * 10   # ns on the 100 MHz clock
        q = lambda x: "min %d p50 %d p90 %d max %d" % (x.min(), np.percentile(x, 50), np.percentile(x, 90), x.max())
        print("   last launch, %d workgroups (ns after the first start): start %s | end %s | duration %s" % (len(blk), q(st), q(en), q(en - st)))
        if len(prev[0]):
            pend = max(prev[0][:, 1].max(), prev[1].max())
            print("   gap: previous launch's last end -> this launch's first start %d ns; previous first start -> this first start %d ns" % ((t0 - pend) * 10, (t0 - prev[0][:, 0].min()) * 10))
        if dst[0] > 0:
            print("   draw wavefront of workgroup 0 (ns after the first start): entry %d, jump entries landed %d, past the barrier %d, raw outputs %d, records stored %d, gathers landed %d, partner2 done %d, stores acknowledged %d" % tuple((dst[[0, 1, 2, 5, 6, 7, 3, 4]] - t0) * 10))
        if dw.max() > 0:
            print("   draw wavefronts end: %s" % q((dw[dw > 0] - t0) * 10))
        if len(blk) >= 8: print("   by XCD (blockIdx %% 8): start p50 %s | end max %s" % ([int(np.percentile(st[k::8], 50)) for k in range(8)], [int(en[k::8].max()) for k in range(8)]))
