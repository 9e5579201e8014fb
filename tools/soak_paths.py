#!/usr/bin/env python3
"""Long-run cross-check: the same chain stepped with one launch per ensemble step and with one launch per half-step
must end in the same state, bit for bit (1 M ensemble steps of the C2 workload each, a few stored steps compared too) --
and, so that this is not purely a comparison of the device with itself, both must agree with the ORACLE over a prefix of
PREFIX ensemble steps (default 10 000: about half a minute of the multi-threaded oracle)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi, workloads

from oracle import pyoracle as po

W, D, STEPS, PREFIX = 16384, 32, int(os.environ.get("STEPS", 1000000)), int(os.environ.get("PREFIX", 10000))
P = workloads.ar1_precision(D, 0.5)
pos = workloads.init_positions(W, D, salt=2)
orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), seed=17)
logp0 = orc.logp(pos)
orc.set_state(pos, logp0)
t0 = time.time()
want_chain, want_acc = orc.run(4, interval=PREFIX // 4, mode=po.MODE_COUNTER, threads=min(16, len(os.sched_getaffinity(0))))
want_state = orc.get_state()
print("oracle: %d steps in %.1f s" % (PREFIX, time.time() - t0), flush=True)
out = []
prefix_ok = True
for full in ("1", "0"):
    os.environ["MCMCPP_HIP_FULL_STEP"] = full
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=17)
    s.set_state(pos, logp0)
    chain, acc = s.run(4, interval=PREFIX // 4)
    ok = np.array_equal(chain, want_chain) and np.array_equal(acc, want_acc) and all(np.array_equal(a, b) for a, b in zip(s.get_state(), want_state))
    print("full_step=%s: the first %d steps equal the oracle's: %s" % (full, PREFIX, ok), flush=True)
    prefix_ok = prefix_ok and ok
    s.set_state(pos, s.calc_logp(pos))
    t0 = time.time()
    chain, _ = s.run(8, interval=STEPS // 8, want_accepted=False)
    dt = time.time() - t0
    st = s.get_state()
    c = s.counters()
    print("full_step=%s: %d steps in %.1f s (%.3e walker-steps/s), accepted %d, near ties %d" % (full, STEPS, dt, W * STEPS / dt, c["accepted"], c["near_ties"]), flush=True)
    out.append((chain, st, c))
    s.close()
same = np.array_equal(out[0][0], out[1][0]) and all(np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1])) and out[0][2]["accepted"] == out[1][2]["accepted"]
print("identical trajectories:", same, "| oracle prefix:", prefix_ok)
sys.exit(0 if same and prefix_ok else 1)
