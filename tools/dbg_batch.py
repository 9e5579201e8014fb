#!/usr/bin/env python3
"""batched draw records: a few steps against the oracle, where the first difference is (experiments)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcpp_amd import capi
from oracle import pyoracle as po
from tests.golden.make_golden import ar1_precision
W, D, steps = 256, 32, 6
P = ar1_precision(D, 0.5, np.float64)
orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), seed=0)
pos = po.init_positions(po.F64, W, D)
lp = orc.logp(pos)
orc.set_state(pos, lp)
want, wacc = orc.run(steps, mode=po.MODE_COUNTER, threads=2)
s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=0)
s.set_state(pos, lp)
got, acc = s.run(steps)
print("accepted", acc, wacc)
for k in range(steps):
    print("step", k, "equal" if np.array_equal(got[k], want[k]) else "DIFFERENT rows %d" % int((got[k] != want[k]).any(axis=1).sum()))
