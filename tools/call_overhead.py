#!/usr/bin/env python3
"""Per-call overhead of mcmcpp_hip_run: many short runs (what a facade user with a per-step PostStepAction causes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi, workloads
W, D = 16384, 32
P = workloads.ar1_precision(D, 0.5)
s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=1)
pos = workloads.init_positions(W, D)
s.set_state(pos, s.calc_logp(pos))
out = np.zeros((1, W, D))
for steps, kw in [(1, dict(save_chain=False, want_accepted=False)), (1, dict(save_chain=True, want_accepted=True, out=out)),
                  (10, dict(save_chain=False, want_accepted=False)), (100, dict(save_chain=False, want_accepted=False))]:
    for _ in range(20):
        s.run(1, interval=steps, **kw)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        s.run(1, interval=steps, **kw)
    dt = (time.perf_counter() - t0) / n
    print("run(1 stored step, interval %3d, %s): %.1f us per call = %.1f us per ensemble step" % (steps, "chain+counters" if kw.get("save_chain") else "nothing stored", dt * 1e6, dt * 1e6 / steps))
