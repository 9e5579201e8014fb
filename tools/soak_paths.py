#!/usr/bin/env python3
"""Long-run cross-check: the same chain stepped with one launch per ensemble step and with one launch per half-step
must end in the same state, bit for bit (1 M ensemble steps of the C2 workload each, a few stored steps compared too)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi, workloads

W, D, STEPS = 16384, 32, int(os.environ.get("STEPS", 1000000))
P = workloads.ar1_precision(D, 0.5)
out = []
for full in ("1", "0"):
    os.environ["MCMCPP_HIP_FULL_STEP"] = full
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=17)
    pos = workloads.init_positions(W, D, salt=2)
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
print("identical trajectories:", same)
sys.exit(0 if same else 1)
