#!/usr/bin/env python3
"""Mover::DifferentialEvolution at the bench's shape (C2: 16384 walkers x 32 dims, dense Gaussian): device walker-steps/s
against the reference's own mover in its sequential sampler (oracle/_ref, one core) and the oracle's restatement.
    python tools/bench_diffevo.py [walkers] [dims] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcpp_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.golden.make_golden import ar1_precision  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    P = ar1_precision(D, 0.5, np.float64).ravel()
    pos = po.init_positions(po.F64, W, D, salt=0)
    if os.environ.get("DE_CALC"):  # (experiments: launch times only, e.g. with the diagnostic knobs that leave the chain wrong)
        iso = os.environ["DE_CALC"] == "iso"
        s = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN if iso else capi.CALC_DENSE_GAUSSIAN, None if iso else P, seed=0, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
        lp = s.calc_logp(pos)
        s.set_state(pos, lp)
        s.run(1, 50, save_chain=False)
        s.run(20, 100, save_chain=False)
        gpu_ms, launches = s.last_run_timing()
        print("%s calculator: %.2f us per half-step" % (os.environ["DE_CALC"], gpu_ms * 1e3 / launches))
        return
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P, seed=0, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = s.calc_logp(pos)
    s.set_state(pos, lp)
    s.run(1, 50, save_chain=False)
    steps = max(100, steps)
    t0 = time.perf_counter()
    _, acc = s.run(steps // 100, 100, save_chain=True)
    dev = time.perf_counter() - t0
    done = (steps // 100) * 100
    c = s.counters()
    gpu_ms, launches = s.last_run_timing()
    # algorithmic bytes of one DifferentialEvolution update (DifferentialEvolution.h:80-112): own row + two partner rows + own
    # log-posterior read, row + log-posterior written = (4D + 2) * 8 bytes (1 040 at D = 32); one launch = one half-step
    bytes_per_launch = (W // 2) * (4 * D + 2) * 8
    us_per_launch = gpu_ms * 1e3 / launches
    achieved = bytes_per_launch / (us_per_launch * 1e-6) / 1e9
    line = "differential evolution %d x %d dense Gaussian: device %.3e walker-steps/s (%.2f us per ensemble step by the wall clock incl. the " \
           "stored steps' download; %.2f us per launch = half-step by HIP events, 2 launches per step), acceptance %.3f, " \
           "%d draws thrown away in %d half-steps; roofline: %.0f GB/s algorithmic = %.3f of 8000 (de_update_kernel + the planning launches, %d B per update)" \
           % (W, D, W * done / dev, dev / done * 1e6, us_per_launch, acc.sum() / (W * done), c["redraws"], 2 * (done + 50), achieved,
              achieved / 8000.0, (4 * D + 2) * 8)
    cpu_steps = max(1, min(30, (30 * 16384) // W))
    orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P, seed=0, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    orc.set_state(pos, lp)
    t0 = time.perf_counter()
    orc.run(cpu_steps, 1, save_chain=False)
    port = W * cpu_steps / (time.perf_counter() - t0)
    line += "; oracle on one core %.3e" % port
    if po.reference_available() and W * D <= 16384 * 32:  # (the reference's Chain overflows its 32-bit indices beyond that)
        r = po.reference_run(W, D, po.CALC_DENSE_GAUSSIAN, P, 0, pos, lp, 1, cpu_steps, want_chain=False, alpha_code=2)
        line += "; reference (one core) %.3e" % (W * cpu_steps / r["seconds"])
    print(line)


if __name__ == "__main__":
    main()
