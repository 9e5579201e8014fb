#!/usr/bin/env python3
"""Row f2 measurement: Analysis::CovarianceMatrix over a C2-sized chain (n stored steps of 16384 x 32 fp64):
the device accumulation (upload from pageable host memory + matrix-core kernel) against the reference's own class
(oracle/_ref, when present) or the oracle's restatement of it, on one host core."""
import json, os, sys, time
import numpy as np
import torch  # (before the library: two HIP runtimes in one process initialise in this order only)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcpp_amd import capi
from oracle import pyoracle as po

n, W, D = int(os.environ.get("STEPS", 20)), 16384, 32
rng = np.random.default_rng(0)
steps = rng.standard_normal((n, W, D)) * np.linspace(0.5, 2.0, D)
m = capi.HipMoments(W, D)
m.add_steps(steps[:2]); m.finish(); m.reset()          # warm-up (allocations, first launches)
t0 = time.perf_counter(); m.add_steps(steps); _, mean, cov, corr = m.finish(); t_dev = time.perf_counter() - t0
dev = torch.from_numpy(steps).cuda()
m.reset(); torch.cuda.synchronize()
t0 = time.perf_counter(); m.add_device_steps(dev.data_ptr(), n); t_res = time.perf_counter() - t0
_, _, cov_res, _ = m.finish()
assert np.allclose(cov_res, cov, rtol=0, atol=1e-12)  # (the upload path adds 64 MiB chunks one after the other)
m.reset(); m.add_steps(steps)
cpu_n = min(n, 4)
t0 = time.perf_counter()
if po.reference_available():
    rcov, rcorr = po.reference_chain_covariance(steps[:cpu_n]); kind = "reference"
else:
    _, rcov, rcorr = po.chain_covariance(steps[:cpu_n]); kind = "port"
t_cpu = time.perf_counter() - t0
m.reset(); m.add_steps(steps[:cpu_n]); _, _, c4, _ = m.finish()
samples = n * W
print(json.dumps({"metric": "chain samples/s through Analysis::CovarianceMatrix, 16384 walkers x 32 dims fp64",
                  "value": samples / t_dev, "unit": "samples/s", "seconds": t_dev, "stored_steps": n,
                  "bytes_uploaded": steps.nbytes, "upload_GBps": steps.nbytes / t_dev / 1e9,
                  "matrix_flops": 2.0 * samples * D * D,
                  "device_resident": {"seconds": t_res, "samples_per_s": samples / t_res, "GBps_read": steps.nbytes / t_res / 1e9,
                                      "hbm_frac_of_8TBps": steps.nbytes / t_res / 8e12}, "max_abs_diff_vs_cpu": float(np.abs(c4 - rcov).max()),
                  "cpu_baseline": {"value": cpu_n * W / t_cpu, "unit": "samples/s", "cores": 1, "kind": kind,
                                   "sample": "%d stored steps, %.2f s" % (cpu_n, t_cpu)}}))
