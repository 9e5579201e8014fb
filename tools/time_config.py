#!/usr/bin/env python3
"""Launch time of one stretch-move configuration: python tools/time_config.py W D calc dtype [batch] [chains]
(calc: iso | dense | rosenbrock; dtype: f64 | f32).  Prints walker-steps/s, the average step-launch time (HIP events on the
launch stream) and the nominal roofline fraction.  Environment knobs apply as usual."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mcmcpp_amd import capi, workloads  # noqa: E402

W, D, calc, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 500
chains = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dt = capi.F32 if dtype == "f32" else capi.F64
elem = 4 if dt == capi.F32 else 8
calc_id, params = {"iso": (capi.CALC_ISO_GAUSSIAN, None), "dense": (capi.CALC_DENSE_GAUSSIAN, workloads.ar1_precision(D, 0.5).ravel()),
                   "rosenbrock": (capi.CALC_ROSENBROCK, [1.0, 100.0, 0.05])}[calc]
s = capi.HipSampler(W, D, calc_id, params, seed=0, dtype=dt, num_chains=chains)
pos = np.stack([workloads.init_positions(W, D, salt=k) for k in range(chains)]) if chains > 1 else workloads.init_positions(W, D, salt=0)
s.set_state(pos, s.calc_logp(pos))
s.run(1, interval=batch, save_chain=False)
gpu_ms, launches, reps, acc = 0.0, 0, 0, 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("TIME_CONFIG_SECONDS", "1.0")):
    _, a = s.run(1, interval=batch, save_chain=False)
    acc += int(a.sum())
    ms, nl = s.last_run_timing()
    gpu_ms += ms
    launches += nl
    reps += 1
el = time.perf_counter() - t0
ws = float(W) * chains * batch * reps
upd = ws / launches
us = gpu_ms * 1e3 / launches
frac = upd * ((2 * D + 1) * elem + (D + 1) * elem) / (us * 1e-6) / 8e12
c = s.counters()
print("%dx%d %s %s chains %d: %.3e walker-steps/s, %.2f us per launch of %d updates, roofline %.3f, acceptance %.4f, near ties %d in %d launches"
      % (W, D, calc, dtype, chains, ws / el, us, upd, frac, acc / ws, c["near_ties"], launches + (2 if upd < 0.75 * W * chains else 1) * batch))
