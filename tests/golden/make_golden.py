#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference; builds oracle/_ref/libmcmcpp_ref.so through
oracle/Makefile).  Every fixture is data: inputs (initial positions / log-probabilities, calculator
parameters, seed) and the reference's outputs (accepted-proposal count after every runMCMC(1) call,
chain steps).  Large cases keep SHA-256 digests of the chain steps instead of the steps themselves.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py iso64x4    # one
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ar1_precision(D, rho, t):
    p = np.zeros((D, D), dtype=t)
    rho = t(rho)
    d = t(1) - rho * rho
    for i in range(D):
        edge = i == 0 or i == D - 1
        p[i, i] = (t(1) if edge else (t(1) + rho * rho)) / d
        if i + 1 < D:
            p[i, i + 1] = -rho / d
            p[i + 1, i] = -rho / d
    return p


CASES = {
    # name: W, D, calc, params-maker, dtype, steps, slicing, chain steps kept in full
    "iso64x4": dict(W=64, D=4, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=1000, keep=[1, 2, 10, 1000]),
    # StretchMove with a non-default stretch scale: GwDistribution<double, 3, 2>
    "iso64x4_alpha3_2": dict(W=64, D=4, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=400, keep=[1, 2, 10, 400],
                             alpha=(3, 2), alpha_code=1),
    "iso100x7": dict(W=100, D=7, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=500, keep=[1, 2, 10, 500]),
    "iso64x4_f32": dict(W=64, D=4, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F32, steps=1000, keep=[1, 2, 10, 1000]),
    "dense96x16": dict(W=96, D=16, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F64, steps=300, keep=[1, 2, 10, 300],
                       rho=0.5),
    "dense80x5_f32": dict(W=80, D=5, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F32, steps=300, keep=[1, 2, 10, 300],
                          rho=0.3),
    "rosen80x8": dict(W=80, D=8, calc=po.CALC_ROSENBROCK, dtype=po.F64, steps=300, keep=[1, 2, 10, 300],
                      params=[1.0, 100.0, 0.05]),
    "rosen200x33": dict(W=200, D=33, calc=po.CALC_ROSENBROCK, dtype=po.F64, steps=100, keep=[1, 100],
                        params=[1.0, 100.0, 0.05]),
    "iso600x130": dict(W=600, D=130, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=40, keep=[],
                       digest=[1, 2, 40], sample_rows=[0, 299, 300, 599]),
    # the reference's own test set-up (test/sequential/SkewedGaussian/StretchMove/src/main.cpp:22-41),
    # shortened to 200 stored steps of slicing 30; Calculator = the reference's SkewedGaussianTwoDim
    "skewed320x2": dict(W=320, D=2, calc=po.CALC_SKEWED_GAUSSIAN_2D, dtype=po.F64, steps=200, slicing=30,
                        keep=[1, 2, 200], params=[0.13], skewed_init=True),
    # BASELINE config 2 (16 384 x 32 correlated Gaussian), 20 steps, digests only
    "c2_16384x32": dict(W=16384, D=32, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F64, steps=20, keep=[], rho=0.5,
                        digest=[1, 2, 20], sample_rows=[0, 1, 8191, 8192, 16383]),
    # BASELINE config 3 shape (Rosenbrock, D = 32) at a reduced walker count, digests only
    # next row f3: Mover::DifferentialEvolution (alpha_code 2 selects it in the reference driver).  W/2 = 50, 40 and 7 are
    # not powers of two (bounded_rand throw-aways) and 7 makes the second partner collide with the first every few updates
    "de_iso64x4": dict(W=64, D=4, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=600, keep=[1, 2, 10, 600], mover=1, alpha_code=2),
    "de_iso100x7": dict(W=100, D=7, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=300, keep=[1, 2, 10, 300], mover=1, alpha_code=2),
    "de_iso14x3": dict(W=14, D=3, calc=po.CALC_ISO_GAUSSIAN, dtype=po.F64, steps=500, keep=[1, 2, 10, 500], mover=1, alpha_code=2),
    "de_rosen80x8": dict(W=80, D=8, calc=po.CALC_ROSENBROCK, dtype=po.F64, steps=300, keep=[1, 2, 10, 300],
                         params=[1.0, 100.0, 0.05], mover=1, alpha_code=2),
    "de_dense96x16": dict(W=96, D=16, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F64, steps=200, keep=[1, 2, 10, 200], rho=0.5, mover=1,
                          alpha_code=2),
    "de_dense80x5_f32": dict(W=80, D=5, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F32, steps=300, keep=[1, 2, 10, 300], rho=0.3, mover=1,
                             alpha_code=2),
    "de_c2_16384x32": dict(W=16384, D=32, calc=po.CALC_DENSE_GAUSSIAN, dtype=po.F64, steps=10, keep=[], rho=0.5, mover=1, alpha_code=2,
                           digest=[1, 2, 10], sample_rows=[0, 1, 8191, 8192, 16383]),
    "c3_4096x32": dict(W=4096, D=32, calc=po.CALC_ROSENBROCK, dtype=po.F64, steps=20, keep=[],
                       params=[1.0, 100.0, 0.05], digest=[1, 2, 20], sample_rows=[0, 2047, 2048, 4095]),
}


def make(name, c):
    W, D, dtype = c["W"], c["D"], c["dtype"]
    t = po.np_dtype(dtype)
    if "rho" in c:
        params = ar1_precision(D, c["rho"], t).ravel()
    elif "params" in c:
        params = np.asarray(c["params"], dtype=t)
    else:
        params = None
    alpha = c.get("alpha", (2, 1))
    orc = po.Oracle(W, D, c["calc"], params, seed=0, dtype=dtype, alpha=alpha, mover=c.get("mover", 0))
    if c.get("skewed_init"):
        pos, logp = po.reference_skewed_initial_values(W, 0.13, 53)
    else:
        pos = po.init_positions(dtype, W, D, salt=0)
        logp = orc.logp(pos)
    slicing = c.get("slicing", 1)
    steps = c["steps"]
    ref = po.reference_run(W, D, c["calc"], params, 0, pos, logp, steps, 1, slicing=slicing, dtype=dtype,
                           alpha_code=c.get("alpha_code", 0))
    assert ref["stored"] == steps + 1 and not ref["chain_full"]
    acc_cum = ref["accepted"].astype(np.int64)
    # reference totals count the initial placement as one accepted step per walker (Walker.h:76,168)
    acc_per_call = np.diff(np.concatenate([[W], acc_cum])).astype(np.uint32)
    out = dict(W=np.int32(W), D=np.int32(D), calc=np.int32(c["calc"]), dtype=np.int32(dtype), seed=np.int64(0),
               slicing=np.int32(slicing), steps=np.int32(steps), alpha=np.asarray(alpha, dtype=np.int32), mover=np.int32(c.get("mover", 0)),
               params=(np.zeros(0, dtype=t) if params is None else params),
               accepted_per_call=acc_per_call, accepted_total=np.uint64(acc_cum[-1]),
               total_steps=np.uint64(ref["total"][-1]))
    big = bool(c.get("digest"))
    if not big:
        out["init_pos"] = pos
        out["init_logp"] = logp
    else:
        out["init_pos_sha256"] = np.array(sha(pos))
        out["init_logp_sha256"] = np.array(sha(logp))
    chain = ref["chain"]
    assert np.array_equal(chain[0], pos)
    for k in c["keep"]:
        out["chain_step_%d" % k] = chain[k]
    for k in c.get("digest", []):
        out["chain_sha256_step_%d" % k] = np.array(sha(chain[k]))
        rows = c.get("sample_rows", [])
        out["chain_rows_step_%d" % k] = chain[k][rows]
    if c.get("sample_rows"):
        out["sample_rows"] = np.asarray(c["sample_rows"], dtype=np.int32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    return dict(name=name, W=W, D=D, steps=steps, slicing=slicing, accepted_total=int(acc_cum[-1]),
                total_steps=int(ref["total"][-1]), bytes=os.path.getsize(path))


def make_reference_test_run():
    """The reference's SkewedGaussian/StretchMove test end to end (320 x 2, slicing 30, 40 019 stored
    steps): its printed acceptance line is 'Acceptance Fraction: 274778560/384182720 | 0.715229'."""
    W, D = 320, 2
    pos, logp = po.reference_skewed_initial_values(W, 0.13, 53)
    ref = po.reference_run(W, D, po.CALC_SKEWED_GAUSSIAN_2D, np.array([0.13]), 0, pos, logp, 1, 40019, slicing=30,
                           want_chain=False)
    out = dict(W=W, D=D, slicing=30, stored_steps=40019, eps=0.13, accepted_total=int(ref["accepted"][-1]),
               total_steps=int(ref["total"][-1]), seconds_reference=ref["seconds"],
               note="reference built with g++ -O2 -ffp-contract=off (oracle/Makefile); a build that lets the "
                    "compiler contract a*b+c into FMA (e.g. -O3 -march=native) takes different accept decisions")
    with open(os.path.join(HERE, "reference_skewed_test.json"), "w") as f:
        json.dump(out, f, indent=1)
    # the same driver with Mover::DifferentialEvolution and slicing 10: test/sequential/SkewedGaussian/DiffEvo/src/main.cpp
    ref = po.reference_run(W, D, po.CALC_SKEWED_GAUSSIAN_2D, np.array([0.13]), 0, pos, logp, 1, 40019, slicing=10,
                           want_chain=False, alpha_code=2)
    de = dict(W=W, D=D, slicing=10, stored_steps=40019, eps=0.13, accepted_total=int(ref["accepted"][-1]),
              total_steps=int(ref["total"][-1]), seconds_reference=ref["seconds"], mover="DifferentialEvolution")
    with open(os.path.join(HERE, "reference_skewed_diffevo_test.json"), "w") as f:
        json.dump(de, f, indent=1)
    return out, de


def make_covariance():
    """Analysis::CovarianceMatrix of the reference (its own Chain, its own iterators) over chains produced by the
    reference's sampler: covariance_<case>.npz holds the chain steps, the slicing and the reference's matrices."""
    out = []
    for name, (src, dtype, first, count, slice_interval) in {
            "covariance_dense96x16": ("dense96x16", po.F64, 60, 48, 1),
            "covariance_dense96x16_slice5": ("dense96x16", po.F64, 20, 121, 5),
            "covariance_rosen80x8": ("rosen80x8", po.F64, 100, 90, 3),
            "covariance_dense80x5_f32": ("dense80x5_f32", po.F32, 50, 100, 2)}.items():
        c = CASES[src]
        t = po.np_dtype(dtype)
        W, D = c["W"], c["D"]
        params = ar1_precision(D, c["rho"], t).ravel() if "rho" in c else (np.array(c["params"], dtype=t) if "params" in c else None)
        pos = po.init_positions(dtype, W, D, salt=3)
        logp = po.Oracle(W, D, c["calc"], params, dtype=dtype).logp(pos)
        ref = po.reference_run(W, D, c["calc"], params, 7, pos, logp, 1, first + count, dtype=dtype)
        steps = np.ascontiguousarray(ref["chain"][first:first + count])
        cov, corr = po.reference_chain_covariance(steps, slice_interval)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, steps=steps, slice_interval=slice_interval, cov=cov, corr=corr)
        out.append(dict(name=name, shape=list(steps.shape), slice_interval=slice_interval, bytes=os.path.getsize(path)))
    return out


AUTOCORR_CLASS = {  # name: (n, W, D, seed, phi, window_scaling, dtype)
    "autocorr_class_f64": (40000, 4, 2, 11, (0.9, 0.5), 4, po.F64),
    "autocorr_class_f32": (36000, 3, 2, 12, (0.8, 0.3), 5, po.F32),
}


def make_autocorr(which=None):
    """Analysis::Detail::AutoCov and Analysis::AutoCorrCalc of the reference.

    autocov_series.npz: series, their averages and the reference's calcNormAutoCov output.
    autocorr_class_*.npz: the reference's calcAutoCorrTimes over a chain from tests/goldens.ar_chain (only its
    parameters and a digest are stored).  The class adds every series onto the scratch array it allocates
    uninitialised (AutoCorrCalc.h:239-245,307-320): its output is only defined when that memory is zero, hence the
    long chains and one class fixture per process (which=...), run with MALLOC_MMAP_THRESHOLD_=65536 so that glibc takes
    the array straight from mmap."""
    from tests.goldens import ar_chain
    out = []
    if which is None:
        data = {}
        for tag, (n, seed, phi, dtype) in {"f64_n100": (100, 1, 0.7, po.F64), "f64_n1000": (1000, 2, 0.95, po.F64), "f64_n1024": (1024, 3, 0.5, po.F64),
                                           "f64_n1025": (1025, 4, 0.9, po.F64), "f64_n3": (3, 5, 0.2, po.F64), "f32_n777": (777, 6, 0.85, po.F32),
                                           "f32_n2048": (2048, 7, 0.6, po.F32)}.items():
            x = ar_chain(n, 1, 1, seed, phi, po.np_dtype(dtype)).ravel()
            avg = po.np_dtype(dtype)(x.astype(np.float64).mean())
            data[tag + "_series"] = x
            data[tag + "_avg"] = np.array(avg)
            data[tag + "_autocov"] = po.reference_norm_autocov(x, float(avg), dtype)
        path = os.path.join(HERE, "autocov_series.npz")
        np.savez_compressed(path, **data)
        out.append(dict(name="autocov_series", bytes=os.path.getsize(path)))
        for name in ["actime_65535", "actime_262143"] + list(AUTOCORR_CLASS):  # one fresh process each
            import subprocess
            # MALLOC_MMAP_THRESHOLD_: glibc then serves every array above 64 KiB from fresh (zero) pages
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "autocorr:" + name], env=dict(os.environ, MALLOC_MMAP_THRESHOLD_="65536"))
        return out
    if which.startswith("actime_"):
        # the reference's own known-answer test of AutoCorrCalc (test/sequential/AcTime/src/main.cpp) run by the reference:
        # 100 walkers, run number 0, the five phi of the test; only the times and a digest of the chain are kept
        n_steps = int(which.split("_")[1])
        chain, times = po.reference_actime_test(n_steps)
        assert np.all(np.isfinite(times)), "the reference's scratch memory was not zero: %r" % (times,)
        path = os.path.join(HERE, which + ".npz")
        np.savez_compressed(path, n_steps=n_steps, W=po.ACTIME_WALKERS, phis=np.array(po.ACTIME_PHIS), chain_sha256=sha(chain), times=times,
                            recorded_in_the_test_source=np.array(po.ACTIME_RECORDED))
        return [dict(name=which, times=times.tolist(), bytes=os.path.getsize(path))]
    n, W, D, seed, phi, scaling, dtype = AUTOCORR_CLASS[which]
    steps = ar_chain(n, W, D, seed, phi, po.np_dtype(dtype))
    times = po.reference_autocorr_times(steps, scaling, dtype)
    assert np.all(np.isfinite(times)), "the reference's scratch memory was not zero: %r" % (times,)
    path = os.path.join(HERE, which + ".npz")
    np.savez_compressed(path, n=n, W=W, D=D, seed=seed, phi=np.array(phi), window_scaling=scaling, dtype=dtype, steps_sha256=sha(steps), times=times)
    return [dict(name=which, times=times.tolist(), bytes=os.path.getsize(path))]


def main():
    want = sys.argv[1:]
    if want and want[0].startswith("autocorr:"):
        print(make_autocorr(want[0].split(":", 1)[1]), flush=True)
        return
    if not po.reference_available():
        sys.exit("oracle/_ref/libmcmcpp_ref.so is not available (no /root/reference here)")
    summary = []
    for name, c in CASES.items():
        if want and name not in want:
            continue
        summary.append(make(name, c))
        print(summary[-1], flush=True)
    if not want or "reference_skewed_test" in want:
        print(make_reference_test_run(), flush=True)
    if not want or "covariance" in want:
        print(make_covariance(), flush=True)
    if not want or "autocorr" in want:
        print(make_autocorr(), flush=True)
    if not want:
        with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
