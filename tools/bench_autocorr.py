#!/usr/bin/env python3
"""Time mcmcpp_hip_autocorr_times on a chain of the bench's shape against the oracle's restatement on one core.
    python tools/bench_autocorr.py [n_steps] [walkers] [dims]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcpp_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    rng = np.random.default_rng(1)
    steps = np.empty((n, W, D))
    steps[0] = rng.standard_normal((W, D))
    phi = np.linspace(0.3, 0.95, D)
    for t in range(1, n):
        steps[t] = phi * steps[t - 1] + rng.standard_normal((W, D))
    capi.autocorr_times(steps[:16], 0, 4)  # library + device warm-up
    t0 = time.perf_counter()
    got = capi.autocorr_times(steps, 0, 4)
    dev = time.perf_counter() - t0
    sub = max(1, min(W, 64))
    t0 = time.perf_counter()
    want_sub = po.autocorr_times(steps[:, :sub, :], 4)
    cpu = (time.perf_counter() - t0) * W / sub
    same = np.array_equal(capi.autocorr_times(steps[:, :sub, :], 0, 4), want_sub)
    print("autocorr %d steps x %d walkers x %d parameters (%.2f GB): device %.3f s (upload included), oracle on one core %.1f s (scaled from %d walkers), "
          "x%.0f; subset bit-identical: %s; tau %.2f..%.2f" % (n, W, D, steps.nbytes / 1e9, dev, cpu, sub, cpu / dev, same, got.min(), got.max()))


if __name__ == "__main__":
    main()
