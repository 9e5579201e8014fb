#!/usr/bin/env python3
"""Long-run cross-check of the three ways a large dense ensemble is stepped by half-steps: the 16-walker matrix-core kernel with
its next draws behind the accept (the default from 49 152 updates per launch on), with the draws in the gather's shadow
(MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS=-1) and the plain kernel (MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS=-1): the same trajectory,
bit for bit, over STEPS ensemble steps of 131 072 x 32 -- and all three equal to the ORACLE over a prefix of PREFIX steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi, workloads
from oracle import pyoracle as po

W, D, STEPS, PREFIX = 131072, 32, int(os.environ.get("STEPS", 200000)), int(os.environ.get("PREFIX", 400))
P = workloads.ar1_precision(D, 0.5)
pos = workloads.init_positions(W, D, salt=4)
orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), seed=23)
logp0 = orc.logp(pos)
orc.set_state(pos, logp0)
t0 = time.time()
want_chain, want_acc = orc.run(2, interval=PREFIX // 2, mode=po.MODE_COUNTER, threads=min(16, len(os.sched_getaffinity(0))))
want_state = orc.get_state()
print("oracle: %d steps of %d x %d in %.1f s" % (PREFIX, W, D, time.time() - t0), flush=True)
finals, all_ok = [], True
for name, env in (("late draws (default)", {}), ("draws in the gather's shadow", {"MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS": "-1"}),
                  ("plain kernel", {"MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS": "-1"})):
    for k in ("MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS", "MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=23)
    s.set_state(pos, logp0)
    chain, acc = s.run(2, interval=PREFIX // 2)
    ok = np.array_equal(chain, want_chain) and np.array_equal(acc, want_acc) and all(np.array_equal(a, b) for a, b in zip(s.get_state(), want_state))
    t0 = time.time()
    chain, _ = s.run(4, interval=STEPS // 4, want_accepted=False)
    dt = time.time() - t0
    st = s.get_state()
    c = s.counters()
    print("%s: first %d steps equal the oracle's: %s; then %d steps in %.1f s (%.3e walker-steps/s), accepted %d, near ties %d"
          % (name, PREFIX, ok, STEPS, dt, W * STEPS / dt, c["accepted"], c["near_ties"]), flush=True)
    finals.append((chain, st))
    all_ok = all_ok and ok
    s.close()
same = all(np.array_equal(finals[0][0], f[0]) and all(np.array_equal(a, b) for a, b in zip(finals[0][1], f[1])) for f in finals[1:])
print("identical trajectories: %s | oracle prefix: %s" % (same, all_ok))
sys.exit(0 if same and all_ok else 1)
