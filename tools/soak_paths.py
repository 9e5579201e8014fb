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


def locate_near_tie():
    """The soak logs a near tie once in 10^12 decisions or so: find the ensemble step it belongs to (the device counter after
    runs from checkpoints, bisected), then repeat exactly that step on the ORACLE from the device's own state in front of it
    (checkpoint = positions + log-posteriors + the step count, mcmcpp_hip_seek / so_seek) and report which decision it was
    and whether both sides took it the same way."""
    os.environ["MCMCPP_HIP_FULL_STEP"] = "1"
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=17)
    lp0 = s.calc_logp(pos)
    s.set_state(pos, lp0)
    done, state = 0, (pos, lp0)
    span = STEPS
    # walk forward in eighths until the counter moves, then halve
    chunk = max(1, STEPS // 8)
    found = None
    while done < STEPS:
        s.run(1, interval=chunk, save_chain=False, want_accepted=False)
        if s.counters()["near_ties"] > 0:
            found = (done, chunk)
            break
        done += chunk
        p_, l_, _ = s.get_state()
        state = (p_, l_)
    if found is None:
        print("no near tie in %d steps" % STEPS)
        return True
    lo, span = found
    while span > 1:
        half = span // 2
        s.set_state(*state)
        s.seek(lo)
        s.run(1, interval=half, save_chain=False, want_accepted=False)
        if s.counters()["near_ties"] > 0:
            span = half
        else:
            p_, l_, _ = s.get_state()
            state = (p_, l_)
            lo, span = lo + half, span - half
    # ensemble step `lo` (counted from 0) holds the near tie; `state` is the ensemble in front of it
    s.set_state(*state)
    s.seek(lo)
    chain, acc = s.run(1)
    o = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), seed=17)
    o.set_state(*state)
    o.seek(lo)
    ochain, oacc = o.run(1, mode=po.MODE_COUNTER, threads=min(16, len(os.sched_getaffinity(0))))
    tie = o.last_near_tie()
    agree = np.array_equal(chain, ochain) and np.array_equal(acc, oacc)
    print("the near tie: ensemble step %d; oracle from the device's state in front of it: near ties %d, %s; the step on the device and on the oracle: %s"
          % (lo, o.near_ties, "half-step %d (colour %s), walker %d, %s, ln U = %.17g against delta = %.17g (margin %.3g)"
             % (tie[0], "black" if tie[0] & 1 else "red", tie[1], "accepted" if tie[2] else "rejected", tie[3], tie[4], abs(tie[3] - tie[4])) if tie else "none recorded",
             "identical" if agree else "DIFFERENT"), flush=True)
    return agree and s.counters()["near_ties"] == o.near_ties


tie_ok = locate_near_tie() if (out[0][2]["near_ties"] or out[1][2]["near_ties"]) else True
same = tie_ok and np.array_equal(out[0][0], out[1][0]) and all(np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1])) and out[0][2]["accepted"] == out[1][2]["accepted"]
print("identical trajectories:", same, "| oracle prefix:", prefix_ok)
sys.exit(0 if same and prefix_ok else 1)
