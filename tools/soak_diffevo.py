#!/usr/bin/env python3
"""Long differential-evolution runs on the device against the oracle, bit for bit (final state, per-step accepted counts,
thrown-away draws): small half-ensembles where the second partner collides often, half-sizes that are not powers of two.
    python tools/soak_diffevo.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcpp_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    bad = 0
    for W, D, calc, params in [(4, 1, po.CALC_ISO_GAUSSIAN, None), (14, 3, po.CALC_ISO_GAUSSIAN, None), (8, 1, po.CALC_ISO_GAUSSIAN, None), (100, 7, po.CALC_ISO_GAUSSIAN, None),
                               (320, 2, po.CALC_SKEWED_GAUSSIAN_2D, [0.13]), (1026, 16, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05]),
                               (4096, 32, po.CALC_ISO_GAUSSIAN, None)]:
        n_steps = steps if W <= 400 else steps // 10
        pos = po.init_positions(po.F64, W, D, salt=9)
        orc = po.Oracle(W, D, calc, params, seed=77, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
        lp = orc.logp(pos)
        orc.set_state(pos, lp)
        _, want_acc = orc.run(n_steps, 1, save_chain=False)
        s = capi.HipSampler(W, D, calc, params, seed=77, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
        s.set_state(pos, lp)
        got_acc = np.concatenate([s.run(n_steps // 4, 1, save_chain=False)[1] for _ in range(4)])
        same = np.array_equal(got_acc, want_acc[:got_acc.size]) if got_acc.size == want_acc.size else False
        for a, b in zip(s.get_state(), orc.get_state()):
            same = same and np.array_equal(a, b)
        c = s.counters()
        same = same and c["redraws"] == orc.redraws
        print("%5d x %2d, %6d steps: %s (draws thrown away %d, oracle %d; near ties %d)" % (W, D, n_steps, "identical" if same else "DIFFERENT", c["redraws"],
                                                                                        orc.redraws, c["near_ties"]), flush=True)
        bad += 0 if same else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
