#!/usr/bin/env python3
"""Long run of the split ensemble over G loop-back ranks (tests/cpp/loopback_ccl.hip) against the SAME ensemble stepped by one
handle: many chunks, the learned block bound adapting, natural overflows (if any) rolled back -- stored steps, final state and
accepted counts must be identical, bit for bit.   python tools/soak_split_loopback.py [G] [W] [D] [steps]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCMCPP_HIP_RCCL_LIB"] = os.path.join(ROOT, "tests", "cpp", "_build", "libloopback_ccl.so")
import numpy as np  # noqa: E402
from mcmcpp_amd import capi, workloads  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
RUNS, INTERVAL = 8, 500
per_run = STEPS // RUNS // INTERVAL
pos = workloads.init_positions(W, D, salt=3)
one = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=4, device=0)
lp = one.calc_logp(pos)
one.set_state(pos, lp)
want = [one.run(per_run, interval=INTERVAL) for _ in range(RUNS)]
want_state = one.get_state()
one.close()
for scheme in ("1", "0"):
    os.environ["MCMCPP_HIP_COMM_FULL_STEP"] = scheme
    cid = capi.comm_unique_id()
    out, errs = [None] * G, []

    def rank_main(r):
        try:
            h = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=4, device=0, comm_world=G, comm_rank=r, comm_id=cid)
            h.set_state(pos, lp)
            got, stats = [], []
            for _ in range(RUNS):
                got.append(h.run(per_run, interval=INTERVAL, save_chain=(r == 0)))
                stats.append(h.last_run_exchange())
            out[r] = (got, h.get_state(), stats)
            h.close()
        except Exception as e:  # noqa: BLE001
            errs.append("rank %d: %s" % (r, e))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    ok = True
    for r in range(G):
        got, state, stats = out[r]
        for k in range(RUNS):
            ok &= np.array_equal(got[k][1], want[k][1])
            if r == 0:
                ok &= np.array_equal(got[k][0], want[k][0])
        ok &= all(np.array_equal(a, b) for a, b in zip(state, want_state))
    stats = out[0][2]
    print("scheme %s, %d ranks, %d x %d, %d steps: identical to one handle: %s | bytes per step first/last run %.0f / %.0f, block slots %s, repeated chunks %s"
          % ("one exchange per step" if scheme == "1" else "one per half-step", G, W, D, per_run * INTERVAL * RUNS, ok, stats[0][0], stats[-1][0],
             [s[2] for s in stats], [s[1] for s in stats]), flush=True)
    if not ok:
        sys.exit(1)
