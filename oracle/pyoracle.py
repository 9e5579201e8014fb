"""ctypes bindings for the CPU checkers under oracle/.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (mcmcpp_amd, libmcmcpp_hip.so) never does.

  Oracle      this repo's C restatement (oracle/liboracle.so, built by oracle/Makefile)
  Reference   the reference itself compiled from /root/reference (oracle/_ref/libmcmcpp_ref.so);
              present only where it was prebuilt in the build container.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

F64, F32 = 0, 1
CALC_ISO_GAUSSIAN, CALC_DENSE_GAUSSIAN, CALC_ROSENBROCK, CALC_SKEWED_GAUSSIAN_2D = 0, 1, 2, 3
MOVER_STRETCH, MOVER_DIFFERENTIAL_EVOLUTION = 0, 1
MODE_SEQUENTIAL, MODE_COUNTER = 0, 1


def np_dtype(dtype):
    return np.float64 if dtype == F64 else np.float32


class _Config(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("num_walkers", C.c_int32), ("num_params", C.c_int32),
                ("calc_id", C.c_int32), ("calc_params", C.c_void_p), ("calc_params_len", C.c_int32),
                ("gw_alpha_num", C.c_int32), ("gw_alpha_den", C.c_int32), ("mover", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_uint64)]


class _Pcg64(C.Structure):
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64),
                ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64)]


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libmcmcpp_ref.so."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("stretch_oracle.c", "stretch_oracle_typed.inc", "stretch_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/MCMCpp/EnsembleSampler.h"):
        for name in ("libmcmcpp_ref.so", "libmcmcpp_ref_o3.so"):
            ref = os.path.join(_HERE, "_ref", name)
            if (force or not os.path.exists(ref)
                    or os.path.getmtime(os.path.join(_HERE, "ref_driver.cpp")) > os.path.getmtime(ref)):
                subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
                break


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.so_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
        L.so_destroy.argtypes = [C.c_void_p]
        L.so_destroy.restype = None
        L.so_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.so_run.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.so_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in ("so_half_steps_done", "so_near_ties", "so_redraws"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_uint64
        L.so_seek.argtypes = [C.c_void_p, C.c_uint64]
        L.so_last_near_tie.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]
        L.so_half_step_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]
        L.so_half_step_commit.argtypes = [C.c_void_p]
        L.so_chain_covariance.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.so_norm_autocov.argtypes = [C.c_int32, C.c_void_p, C.c_double, C.c_int32]
        L.so_autocorr_times.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.so_ar1_test_chain.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.so_positions_ptr.argtypes = [C.c_void_p]
        L.so_positions_ptr.restype = C.c_void_p
        L.so_logp_ptr.argtypes = [C.c_void_p]
        L.so_logp_ptr.restype = C.c_void_p
        L.so_calc_logp.argtypes = [C.POINTER(_Config), C.c_void_p, C.c_void_p]
        L.so_pcg64_seed.argtypes = [C.POINTER(_Pcg64), C.c_uint64, C.c_uint64]
        L.so_pcg64_seed.restype = None
        L.so_pcg64_next.argtypes = [C.POINTER(_Pcg64)]
        L.so_pcg64_next.restype = C.c_uint64
        L.so_pcg64_advance.argtypes = [C.POINTER(_Pcg64), C.c_uint64, C.c_uint64]
        L.so_pcg64_advance.restype = None
        L.so_pcg64_jump_coeffs.argtypes = [C.c_uint64] * 4 + [C.POINTER(C.c_uint64 * 2)] * 2
        L.so_pcg64_jump_coeffs.restype = None
        L.so_canonical_f64.argtypes = [C.c_uint64]
        L.so_canonical_f64.restype = C.c_double
        L.so_canonical_f32.argtypes = [C.c_uint64]
        L.so_canonical_f32.restype = C.c_float
        L.so_init_positions.argtypes = [C.c_int32, C.c_int64, C.c_uint64, C.c_void_p]
        L.so_init_positions.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Pcg64:
    """pcg64 engine of the oracle (known-answer tests)."""

    def __init__(self, seed, stream=0):
        self.g = _Pcg64()
        lib().so_pcg64_seed(C.byref(self.g), seed & (2**64 - 1), stream & (2**64 - 1))

    def next(self):
        return lib().so_pcg64_next(C.byref(self.g))

    def advance(self, delta):
        lib().so_pcg64_advance(C.byref(self.g), (delta >> 64) & (2**64 - 1), delta & (2**64 - 1))

    @property
    def state(self):
        return (self.g.state_hi << 64) | self.g.state_lo

    @property
    def inc(self):
        return (self.g.inc_hi << 64) | self.g.inc_lo


def jump_coeffs(inc, delta):
    m = (C.c_uint64 * 2)()
    p = (C.c_uint64 * 2)()
    M = 2**64 - 1
    lib().so_pcg64_jump_coeffs((inc >> 64) & M, inc & M, (delta >> 64) & M, delta & M, C.byref(m), C.byref(p))
    return (m[0] << 64) | m[1], (p[0] << 64) | p[1]


def init_positions(dtype, W, D, salt=0):
    out = np.empty((W, D), dtype=np_dtype(dtype))
    lib().so_init_positions(dtype, W * D, salt, _ptr(out))
    return out


class Oracle:
    """The CPU restatement of EnsembleSampler + StretchMove."""

    def __init__(self, W, D, calc_id, params=None, seed=0, stream=0, dtype=F64, alpha=(2, 1), mover=0):
        self.W, self.D, self.dtype = W, D, dtype
        self.np_t = np_dtype(dtype)
        self.params = None if params is None else np.ascontiguousarray(params, dtype=self.np_t).ravel()
        self.cfg = _Config(dtype, W, D, calc_id, _ptr(self.params), 0 if self.params is None else self.params.size,
                           alpha[0], alpha[1], mover, seed & (2**64 - 1), stream & (2**64 - 1))
        self.h = C.c_void_p()
        rc = lib().so_create(C.byref(self.cfg), C.byref(self.h))
        if rc:
            raise ValueError("so_create failed: %d" % rc)

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_destroy(self.h)
            self.h = None

    def logp(self, pos):
        pos = np.ascontiguousarray(pos, dtype=self.np_t).reshape(-1, self.D)
        out = np.empty(pos.shape[0], dtype=self.np_t)
        for i in range(pos.shape[0]):
            rc = lib().so_calc_logp(C.byref(self.cfg), _ptr(pos[i]), out[i:i + 1].ctypes.data_as(C.c_void_p))
            assert rc == 0
        return out

    def set_state(self, pos, logp):
        pos = np.ascontiguousarray(pos, dtype=self.np_t)
        logp = np.ascontiguousarray(logp, dtype=self.np_t)
        assert pos.size == self.W * self.D and logp.size == self.W
        assert lib().so_set_state(self.h, _ptr(pos), _ptr(logp)) == 0

    def run(self, n_saved, interval=1, save_chain=True, mode=MODE_SEQUENTIAL, threads=1):
        chain = np.empty((n_saved, self.W, self.D), dtype=self.np_t) if save_chain else None
        acc = np.zeros(n_saved * interval, dtype=np.uint32)
        rc = lib().so_run(self.h, n_saved, interval, _ptr(chain), _ptr(acc), mode, threads)
        if rc:
            raise ValueError("so_run failed: %d" % rc)
        return chain, acc

    def get_state(self):
        pos = np.empty((self.W, self.D), dtype=self.np_t)
        logp = np.empty(self.W, dtype=self.np_t)
        nacc = np.empty(self.W, dtype=np.uint32)
        assert lib().so_get_state(self.h, _ptr(pos), _ptr(logp), _ptr(nacc)) == 0
        return pos, logp, nacc

    def half_step_shard(self, color, begin, count):
        a = C.c_uint32(0)
        rc = lib().so_half_step_shard(self.h, color, begin, count, C.byref(a))
        if rc:
            raise ValueError("so_half_step_shard failed: %d" % rc)
        return a.value

    def half_step_commit(self):
        assert lib().so_half_step_commit(self.h) == 0

    def positions_view(self):
        """numpy view (no copy) of the sampler's own [W, D] position array."""
        buf = (C.c_char * (self.W * self.D * np.dtype(self.np_t).itemsize)).from_address(lib().so_positions_ptr(self.h))
        return np.frombuffer(buf, dtype=self.np_t).reshape(self.W, self.D)

    def logp_view(self):
        buf = (C.c_char * (self.W * np.dtype(self.np_t).itemsize)).from_address(lib().so_logp_ptr(self.h))
        return np.frombuffer(buf, dtype=self.np_t)

    def seek(self, ensemble_steps_done):
        """Position the random stream as if that many ensemble steps had been executed since set_state (StretchMove)."""
        assert lib().so_seek(self.h, ensemble_steps_done) == 0

    def last_near_tie(self):
        """(half-step, walker, accepted, ln U, delta) of the most recent near tie, or None."""
        hs, w, a, u, d = C.c_uint64(0), C.c_uint32(0), C.c_int32(0), C.c_double(0), C.c_double(0)
        if lib().so_last_near_tie(self.h, C.byref(hs), C.byref(w), C.byref(a), C.byref(u), C.byref(d)) != 0:
            return None
        return hs.value, w.value, bool(a.value), u.value, d.value

    @property
    def near_ties(self):
        return lib().so_near_ties(self.h)

    @property
    def redraws(self):
        return lib().so_redraws(self.h)


# ---- the reference itself (only where oracle/_ref was prebuilt) ------------------------------------

_ref = None


def reference_available():
    build()
    return os.path.exists(os.path.join(_HERE, "_ref", "libmcmcpp_ref.so"))


_ref_timed = None


def ref_lib(timed_build=False):
    """timed_build: the reference at its own optimisation level (oracle/Makefile: -O3, contraction on) -- for timing
    only; parity work uses the contraction-off build."""
    global _ref, _ref_timed
    if timed_build:
        if _ref_timed is None:
            build()
            path = os.path.join(_HERE, "_ref", "libmcmcpp_ref_o3.so")
            R = C.CDLL(path)
            R.ref_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
            _ref_timed = R
        return _ref_timed
    if _ref is None:
        build()
        R = C.CDLL(os.path.join(_HERE, "_ref", "libmcmcpp_ref.so"))
        R.ref_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p,
                              C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        R.ref_chain_covariance.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_norm_autocov.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_int]
        R.ref_autocorr_times.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
        R.ref_actime_test.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        R.ref_skewed_initial_values.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int]
        R.ref_skewed_initial_values.restype = None
        _ref = R
    return _ref


def timed_reference_available():
    build()
    return os.path.exists(os.path.join(_HERE, "_ref", "libmcmcpp_ref_o3.so"))


def reference_run(W, D, calc_id, params, seed, pos, logp, n_calls, steps_per_call, slicing=1, want_chain=True,
                  threads=0, dtype=F64, alpha_code=0, timed_build=False):
    """Run MCMC::EnsembleSampler (threads=0) or ParallelEnsembleSampler (threads>=1) of the reference.
    alpha_code 0: StretchMove's default GwDistribution<T,2,1>; 1: GwDistribution<T,3,2>; 2: Mover::DifferentialEvolution.

    Returns dict(chain[(stored, W, D)] incl. step 0 = initial placement, accepted[n_calls], total[n_calls],
    stored, seconds, fraction)."""
    t = np_dtype(dtype)
    pos = np.ascontiguousarray(pos, dtype=t)
    logp = np.ascontiguousarray(logp, dtype=t)
    prm = None if params is None else np.ascontiguousarray(params, dtype=t).ravel()
    cap = 1 + n_calls * steps_per_call if want_chain else 0
    chain = np.zeros((cap, W, D), dtype=t) if want_chain else None
    acc = np.zeros(n_calls, dtype=np.uint64)
    tot = np.zeros(n_calls, dtype=np.uint64)
    stored = C.c_int(0)
    secs = C.c_double(0)
    frac = C.c_double(0)
    rc = ref_lib(timed_build).ref_run(dtype, threads, alpha_code, W, D, calc_id, _ptr(prm), seed, _ptr(pos), _ptr(logp), n_calls,
                           steps_per_call, slicing, _ptr(chain), cap, _ptr(acc), _ptr(tot), C.byref(stored),
                           C.byref(secs), C.byref(frac))
    if rc < 0:
        raise ValueError("ref_run failed: %d" % rc)
    return dict(chain=chain, accepted=acc, total=tot, stored=stored.value, seconds=secs.value,
                fraction=frac.value, chain_full=(rc == 1))


def chain_covariance(steps, slice_interval=1):
    """Analysis::CovarianceMatrix restated (oracle): steps[(n, W, D)] -> (mean[D], cov[D, D], corr[D, D])."""
    steps = np.ascontiguousarray(steps)
    dtype = F64 if steps.dtype == np.float64 else F32
    n, W, D = steps.shape
    mean = np.zeros(D, steps.dtype)
    cov = np.zeros((D, D), steps.dtype)
    corr = np.zeros((D, D), steps.dtype)
    rc = lib().so_chain_covariance(dtype, _ptr(steps), n, W, D, slice_interval, _ptr(mean), _ptr(cov), _ptr(corr))
    if rc:
        raise ValueError("so_chain_covariance failed: %d" % rc)
    return mean, cov, corr


def norm_autocov(series, avg, dtype=F64):
    """Restatement of Analysis::Detail::AutoCov::calcNormAutoCov on one series."""
    x = np.array(series, dtype=np_dtype(dtype))
    rc = lib().so_norm_autocov(dtype, _ptr(x), float(avg), x.size)
    if rc:
        raise ValueError("so_norm_autocov failed: %d" % rc)
    return x


def autocorr_times(steps, window_scaling=4, emulate_defect=False, dtype=F64, want_functions=False):
    """Restatement of Analysis::AutoCorrCalc::calcAutoCorrTimes (all walkers) over steps[n][W][D] -> [D]
    (with want_functions: also the averaged autocovariance functions [D][n])."""
    steps = np.ascontiguousarray(steps, dtype=np_dtype(dtype))
    n, W, D = steps.shape
    out = np.zeros(D, steps.dtype)
    fn = np.zeros((D, n), steps.dtype) if want_functions else None
    rc = lib().so_autocorr_times(dtype, _ptr(steps), n, W, D, window_scaling, int(bool(emulate_defect)), _ptr(out), _ptr(fn))
    if rc:
        raise ValueError("so_autocorr_times failed: %d" % rc)
    return (out, fn) if want_functions else out


def reference_chain_covariance(steps, slice_interval=1):
    """The reference's own Analysis::CovarianceMatrix over its own Chain holding `steps`: (cov, corr)."""
    steps = np.ascontiguousarray(steps)
    dtype = F64 if steps.dtype == np.float64 else F32
    n, W, D = steps.shape
    cov = np.zeros((D, D), steps.dtype)
    corr = np.zeros((D, D), steps.dtype)
    rc = ref_lib().ref_chain_covariance(dtype, _ptr(steps), n, W, D, slice_interval, _ptr(cov), _ptr(corr))
    if rc:
        raise ValueError("ref_chain_covariance failed: %d" % rc)
    return cov, corr


def reference_norm_autocov(series, avg, dtype=F64):
    """Analysis::Detail::AutoCov::calcNormAutoCov of the reference on one series."""
    x = np.array(series, dtype=np_dtype(dtype))
    ref_lib().ref_norm_autocov(dtype, _ptr(x), float(avg), x.size)
    return x


def reference_autocorr_times(steps, window_scaling=4, dtype=F64):
    """Analysis::AutoCorrCalc::calcAutoCorrTimes of the reference (all walkers) over steps[n][W][D]."""
    steps = np.ascontiguousarray(steps, dtype=np_dtype(dtype))
    n, W, D = steps.shape
    out = np.zeros(D, steps.dtype)
    rc = ref_lib().ref_autocorr_times(dtype, _ptr(steps), n, W, D, window_scaling, _ptr(out))
    if rc:
        raise ValueError("ref_autocorr_times failed: %d" % rc)
    return out


# the reference's AutoCorrCalc known-answer test (test/sequential/AcTime/src/main.cpp:23-34): five AR(1) parameters with
# analytic times 9, 20, 30, 60, 200; 100 walkers, 262143 steps, run number 0; its recorded outputs (main.cpp:16-22)
ACTIME_PHIS = (0.8, 0.904761904762, 0.9354838709677, 0.9672131147541, 0.990050200903734685)
ACTIME_RECORDED = (9.01951, 19.9437, 29.7831, 59.8488, 196.85)
ACTIME_WALKERS, ACTIME_STEPS = 100, 262143


def ar1_test_chain(n_steps, W=ACTIME_WALKERS, phis=ACTIME_PHIS, run_number=0):
    """Restatement of the chain of the reference's AcTime test: [n_steps + 1][W][D] (offsets 1, variances 1)."""
    D = len(phis)
    chain = np.empty((n_steps + 1, W, D), dtype=np.float64)
    ones = np.ones(D)
    ph = np.array(phis, dtype=np.float64)
    rc = lib().so_ar1_test_chain(run_number, W, D, n_steps, _ptr(ones), _ptr(ph), _ptr(ones), _ptr(chain))
    if rc:
        raise ValueError("so_ar1_test_chain failed: %d" % rc)
    return chain


def reference_actime_test(n_steps, W=ACTIME_WALKERS, phis=ACTIME_PHIS, run_number=0, want_chain=True):
    """The reference's AcTime test run by the reference itself: (chain or None, times[D])."""
    D = len(phis)
    chain = np.empty((n_steps + 1, W, D), dtype=np.float64) if want_chain else None
    times = np.zeros(D)
    ones = np.ones(D)
    ph = np.array(phis, dtype=np.float64)
    stored = ref_lib().ref_actime_test(run_number, W, D, n_steps, _ptr(ones), _ptr(ph), _ptr(ones), _ptr(chain), _ptr(times))
    assert stored == n_steps + 1, stored
    return chain, times


def reference_skewed_initial_values(W=320, eps=0.13, extra_run_number=53):
    init = np.empty((W, 2), dtype=np.float64)
    aux = np.empty(W, dtype=np.float64)
    ref_lib().ref_skewed_initial_values(_ptr(init), _ptr(aux), W, eps, extra_run_number)
    return init, aux
