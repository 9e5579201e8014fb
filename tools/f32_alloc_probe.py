#!/usr/bin/env python3
"""Is the launch time of an fp32 ensemble a property of where its buffers landed?  Creates the same sampler several times in
one process (and is meant to be started several times) and prints the launch time beside the device address of the positions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mcmcpp_amd import capi, workloads
W, D = int(sys.argv[1]), int(sys.argv[2])
dt = capi.F32 if sys.argv[3] == "f32" else capi.F64
keep = []
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 4):
    s = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=0, dtype=dt)
    pos = workloads.init_positions(W, D, salt=0)
    s.set_state(pos, s.calc_logp(pos))
    s.run(1, interval=200, save_chain=False)
    ms = n = 0
    for _ in range(5):
        s.run(1, interval=200, save_chain=False)
        a, b = s.last_run_timing(); ms += a; n += b
    print("rep %d: positions at 0x%x (mod 2 MiB: 0x%x), %.2f us per launch" % (rep, s.device_positions(), s.device_positions() % (2 << 20), ms * 1e3 / n), flush=True)
    if rep % 2 == 0:
        keep.append(s)   # (keep some alive so that the next allocation lands elsewhere)
    else:
        s.close()
