"""ctypes binding of include/mcmcpp_hip.h.  There is no CPU fallback: if libmcmcpp_hip.so cannot be
loaded, or a call fails, an exception carrying the library's own error message is raised."""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

F64, F32 = 0, 1
CALC_ISO_GAUSSIAN, CALC_DENSE_GAUSSIAN, CALC_ROSENBROCK, CALC_SKEWED_GAUSSIAN_2D = 0, 1, 2, 3
MOVER_STRETCH, MOVER_DIFFERENTIAL_EVOLUTION = 0, 1
OK = 0

# every symbol include/mcmcpp_hip.h declares
EXPORTS = [
    "mcmcpp_hip_abi_version", "mcmcpp_hip_register_calculator", "mcmcpp_hip_create", "mcmcpp_hip_destroy", "mcmcpp_hip_last_error",
    "mcmcpp_hip_set_state", "mcmcpp_hip_seek", "mcmcpp_hip_run", "mcmcpp_hip_get_state", "mcmcpp_hip_reset_counters",
    "mcmcpp_hip_get_counters", "mcmcpp_hip_calc_logp", "mcmcpp_hip_last_run_timing", "mcmcpp_hip_last_run_host_timing",
    "mcmcpp_hip_comm_unique_id", "mcmcpp_hip_last_run_exchange", "mcmcpp_hip_run_async", "mcmcpp_hip_wait_stored", "mcmcpp_hip_run_wait",
    "mcmcpp_hip_host_alloc", "mcmcpp_hip_host_free",
    "mcmcpp_hip_half_step_async", "mcmcpp_hip_bind_device_chain", "mcmcpp_hip_device_positions",
    "mcmcpp_hip_shard_span", "mcmcpp_hip_synchronize",
    "mcmcpp_hip_moments_create", "mcmcpp_hip_moments_destroy", "mcmcpp_hip_moments_reset", "mcmcpp_hip_moments_add_steps",
    "mcmcpp_hip_moments_add_device_steps", "mcmcpp_hip_moments_finish", "mcmcpp_hip_moments_last_error",
    "mcmcpp_hip_autocorr_times", "mcmcpp_hip_autocorr_times_device", "mcmcpp_hip_autocorr_last_error",
]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dtype", C.c_int32), ("num_walkers", C.c_int32),
                ("num_params", C.c_int32), ("calc_id", C.c_int32), ("calc_params_len", C.c_int32),
                ("calc_params", C.c_void_p), ("seed", C.c_uint64), ("stream", C.c_uint64), ("device", C.c_int32),
                ("shard_begin", C.c_int32), ("shard_count", C.c_int32), ("graph_steps", C.c_int32),
                ("gw_alpha_num", C.c_int32), ("gw_alpha_den", C.c_int32),
                ("device_positions", C.c_void_p), ("hip_stream", C.c_void_p), ("flags", C.c_uint32),
                ("mover", C.c_uint32), ("comm_world", C.c_int32), ("comm_rank", C.c_int32), ("comm_id", C.c_void_p),
                ("comm", C.c_void_p), ("num_chains", C.c_int32), ("reserved0", C.c_int32)]

FLAG_CALLER_STREAM = 1
COMM_ID_BYTES = 128


def library_path():
    # MCMCPP_HIP_LIB: diagnostics only (e.g. the in-kernel-stamp build libmcmcpp_hip_stamps.so)
    return os.environ.get("MCMCPP_HIP_LIB") or os.path.join(_HERE, "libmcmcpp_hip.so")


def build_library(force=False):
    """Compile every HIP source for gfx950 into mcmcpp_amd/libmcmcpp_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)" % path)
        L = C.CDLL(path)
        vp, i32, i64, u64p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_uint64)
        L.mcmcpp_hip_abi_version.restype = C.c_int
        L.mcmcpp_hip_register_calculator.argtypes = [i32, vp, vp, i32]
        L.mcmcpp_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.mcmcpp_hip_destroy.argtypes = [vp]
        L.mcmcpp_hip_destroy.restype = None
        L.mcmcpp_hip_last_error.argtypes = [vp]
        L.mcmcpp_hip_last_error.restype = C.c_char_p
        L.mcmcpp_hip_set_state.argtypes = [vp, vp, vp]
        L.mcmcpp_hip_seek.argtypes = [vp, C.c_uint64]
        L.mcmcpp_hip_run.argtypes = [vp, i64, i32, vp, vp]
        L.mcmcpp_hip_get_state.argtypes = [vp, vp, vp, vp]
        L.mcmcpp_hip_reset_counters.argtypes = [vp]
        L.mcmcpp_hip_get_counters.argtypes = [vp, u64p, u64p, u64p, u64p]
        L.mcmcpp_hip_calc_logp.argtypes = [vp, vp, i64, vp]
        L.mcmcpp_hip_last_run_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
        dp = C.POINTER(C.c_double)
        L.mcmcpp_hip_last_run_host_timing.argtypes = [vp, dp, dp, dp]
        L.mcmcpp_hip_comm_unique_id.argtypes = [vp]
        L.mcmcpp_hip_last_run_exchange.argtypes = [vp, dp, C.POINTER(i64), C.POINTER(i64)]
        L.mcmcpp_hip_run_async.argtypes = [vp, i64, i32, vp, vp]
        L.mcmcpp_hip_wait_stored.argtypes = [vp, i64]
        L.mcmcpp_hip_run_wait.argtypes = [vp]
        L.mcmcpp_hip_host_alloc.argtypes = [C.c_uint64]
        L.mcmcpp_hip_host_alloc.restype = vp
        L.mcmcpp_hip_host_free.argtypes = [vp]
        L.mcmcpp_hip_host_free.restype = None
        L.mcmcpp_hip_half_step_async.argtypes = [vp, i32, i64]
        L.mcmcpp_hip_bind_device_chain.argtypes = [vp, vp, i64]
        L.mcmcpp_hip_device_positions.argtypes = [vp]
        L.mcmcpp_hip_device_positions.restype = vp
        L.mcmcpp_hip_shard_span.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64)]
        L.mcmcpp_hip_synchronize.argtypes = [vp]
        if hasattr(L, "mcmcpp_hip_moments_create"):  # (absent only from older experiment builds selected with MCMCPP_HIP_LIB)
            L.mcmcpp_hip_moments_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
            L.mcmcpp_hip_moments_destroy.argtypes = [vp]
            L.mcmcpp_hip_moments_destroy.restype = None
            L.mcmcpp_hip_moments_reset.argtypes = [vp]
            L.mcmcpp_hip_moments_add_steps.argtypes = [vp, vp, i64, i64]
            L.mcmcpp_hip_moments_add_device_steps.argtypes = [vp, vp, i64]
            L.mcmcpp_hip_moments_finish.argtypes = [vp, C.POINTER(i64), vp, vp, vp]
            L.mcmcpp_hip_moments_last_error.argtypes = [vp]
            L.mcmcpp_hip_moments_last_error.restype = C.c_char_p
        if hasattr(L, "mcmcpp_hip_autocorr_times"):
            L.mcmcpp_hip_autocorr_times.argtypes = [i32, i32, C.POINTER(vp), i64, i32, i32, i32, i32, vp, vp]
            L.mcmcpp_hip_autocorr_times_device.argtypes = [i32, i32, vp, i64, i32, i32, i32, i32, vp, vp]
            L.mcmcpp_hip_autocorr_last_error.argtypes = []
            L.mcmcpp_hip_autocorr_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def comm_unique_id():
    """The rendezvous token of a new RCCL communicator (make it on one rank, hand it to all)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().mcmcpp_hip_comm_unique_id(buf)
    if rc != OK:
        raise HipError(rc, lib().mcmcpp_hip_last_error(None).decode())
    return buf.raw


def pinned_empty(shape, dtype=np.float64):
    """A numpy array in pinned host memory (mcmcpp_hip_host_alloc), e.g. a chain block the launches can write into."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    p = lib().mcmcpp_hip_host_alloc(n * dt.itemsize)
    if not p:
        raise MemoryError("mcmcpp_hip_host_alloc(%d) failed" % (n * dt.itemsize))
    buf = (C.c_char * (n * dt.itemsize)).from_address(p)
    # the memory goes back when nothing refers to the buffer any more (numpy keeps it alive through arr.base)
    weakref.finalize(buf, lib().mcmcpp_hip_host_free, p)
    return np.frombuffer(buf, dtype=dt).reshape(shape)


def np_dtype(dtype):
    return np.float64 if dtype == F64 else np.float32


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "mcmcpp_hip error %d: %s" % (code, msg))
        self.code = code


class HipSampler:
    """Thin owner of one mcmcpp_hip_sampler handle (one GPU)."""

    def __init__(self, W, D, calc_id, params=None, seed=0, stream=0, dtype=F64, device=-1, shard_begin=0,
                 shard_count=0, graph_steps=0, device_positions=None, hip_stream=None, alpha=(2, 1), mover=0,
                 comm_world=0, comm_rank=0, comm_id=None, comm=None, num_chains=0):
        """comm_world >= 1: this handle is rank comm_rank of a split ensemble (comm_id: the 128 bytes of comm_unique_id(),
        the same on all ranks; or comm: an existing ncclComm_t as an integer); run() then steps the split ensemble."""
        self.W, self.D, self.dtype = W, D, dtype
        self.K = num_chains if num_chains > 1 else 1  # independent ensembles stepped together (leading array dimension)
        self.np_t = np_dtype(dtype)
        self.params = None if params is None else np.ascontiguousarray(params, dtype=self.np_t).ravel()
        self._comm_id = None if comm_id is None else C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self.cfg = Config(C.sizeof(Config), dtype, W, D, calc_id, 0 if self.params is None else self.params.size,
                          _ptr(self.params), seed & (2**64 - 1), stream & (2**64 - 1), device, shard_begin,
                          shard_count, graph_steps, alpha[0], alpha[1], device_positions,
                          0 if hip_stream is None else hip_stream, 0 if hip_stream is None else FLAG_CALLER_STREAM, mover,
                          comm_world, comm_rank, None if self._comm_id is None else C.cast(self._comm_id, C.c_void_p), comm,
                          num_chains, 0)
        self.h = C.c_void_p()
        rc = lib().mcmcpp_hip_create(C.byref(self.cfg), C.byref(self.h))
        if rc != OK:
            raise HipError(rc, lib().mcmcpp_hip_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            lib().mcmcpp_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != OK:
            raise HipError(rc, lib().mcmcpp_hip_last_error(self.h).decode())

    def set_state(self, pos, logp):
        pos = np.ascontiguousarray(pos, dtype=self.np_t)
        logp = np.ascontiguousarray(logp, dtype=self.np_t)
        assert pos.size == self.K * self.W * self.D and logp.size == self.K * self.W
        self._check(lib().mcmcpp_hip_set_state(self.h, _ptr(pos), _ptr(logp)))

    def seek(self, ensemble_steps_done):
        self._check(lib().mcmcpp_hip_seek(self.h, ensemble_steps_done))

    def run(self, n_saved, interval=1, save_chain=True, want_accepted=True, out=None):
        """out: optional preallocated (n_saved, W, D) array receiving the stored steps (like a Chain block that
        already exists); by default a fresh array is allocated."""
        lead = (self.K,) if self.K > 1 else ()  # several chains: chain[K][n_saved][W][D], acc[K][steps]
        if out is not None:
            assert save_chain and out.shape == lead + (n_saved, self.W, self.D) and out.dtype == self.np_t and out.flags.c_contiguous
        chain = out if out is not None else (np.empty(lead + (n_saved, self.W, self.D), dtype=self.np_t) if save_chain else None)
        acc = np.zeros(lead + (n_saved * interval,), dtype=np.uint32) if want_accepted else None
        self._check(lib().mcmcpp_hip_run(self.h, n_saved, interval, _ptr(chain), _ptr(acc)))
        return chain, acc

    def run_async(self, n_saved, interval=1, out=None, want_accepted=False):
        """Start the run on the handle's worker thread; returns (chain, acc) arrays that fill up as it proceeds."""
        chain = out if out is not None else np.empty((n_saved, self.W, self.D), dtype=self.np_t)
        acc = np.zeros(n_saved * interval, dtype=np.uint32) if want_accepted else None
        self._async_keep = (chain, acc)
        self._check(lib().mcmcpp_hip_run_async(self.h, n_saved, interval, _ptr(chain), _ptr(acc)))
        return chain, acc

    def wait_stored(self, count):
        self._check(lib().mcmcpp_hip_wait_stored(self.h, count))

    def run_wait(self):
        self._check(lib().mcmcpp_hip_run_wait(self.h))

    def get_state(self):
        lead = (self.K,) if self.K > 1 else ()
        pos = np.empty(lead + (self.W, self.D), dtype=self.np_t)
        logp = np.empty(lead + (self.W,), dtype=self.np_t)
        nacc = np.empty(lead + (self.W,), dtype=np.uint32)
        self._check(lib().mcmcpp_hip_get_state(self.h, _ptr(pos), _ptr(logp), _ptr(nacc)))
        return pos, logp, nacc

    def reset_counters(self):
        self._check(lib().mcmcpp_hip_reset_counters(self.h))

    def counters(self):
        v = [C.c_uint64(0) for _ in range(4)]
        self._check(lib().mcmcpp_hip_get_counters(self.h, *[C.byref(x) for x in v]))
        return dict(accepted=v[0].value, ensemble_steps=v[1].value, near_ties=v[2].value, redraws=v[3].value)

    def calc_logp(self, pos):
        pos = np.ascontiguousarray(pos, dtype=self.np_t).reshape(-1, self.D)
        out = np.empty(pos.shape[0], dtype=self.np_t)
        self._check(lib().mcmcpp_hip_calc_logp(self.h, _ptr(pos), pos.shape[0], _ptr(out)))
        return out

    def last_run_timing(self):
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(lib().mcmcpp_hip_last_run_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_run_host_timing(self):
        """(ms the host spent enqueueing, wall ms of the call, GPU us of one step's exchange -- split ensembles only)"""
        v = [C.c_double(0) for _ in range(3)]
        self._check(lib().mcmcpp_hip_last_run_host_timing(self.h, *[C.byref(x) for x in v]))
        return v[0].value, v[1].value, v[2].value

    def last_run_exchange(self):
        """Split ensembles: (bytes this rank received per ensemble step, chunks repeated with larger blocks, slots per block)"""
        b, r, c = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._check(lib().mcmcpp_hip_last_run_exchange(self.h, C.byref(b), C.byref(r), C.byref(c)))
        return b.value, r.value, c.value

    def half_step_async(self, color, save_slot=-1):
        self._check(lib().mcmcpp_hip_half_step_async(self.h, color, save_slot))

    def bind_device_chain(self, ptr, slots):
        self._check(lib().mcmcpp_hip_bind_device_chain(self.h, ptr, slots))

    def device_positions(self):
        return lib().mcmcpp_hip_device_positions(self.h)

    def shard_span(self, color):
        off, cnt = C.c_int64(0), C.c_int64(0)
        self._check(lib().mcmcpp_hip_shard_span(self.h, color, C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def synchronize(self):
        self._check(lib().mcmcpp_hip_synchronize(self.h))


class HipMoments:
    """Device-side Analysis::CovarianceMatrix (include/mcmcpp_hip.h, mcmcpp_hip_moments_*)."""

    def __init__(self, num_walkers, num_params, dtype=F64, device=-1):
        self.W, self.D, self.dtype = num_walkers, num_params, dtype
        self.h = C.c_void_p()
        rc = lib().mcmcpp_hip_moments_create(dtype, device, num_walkers, num_params, C.byref(self.h))
        if rc != OK:
            raise HipError(rc, (lib().mcmcpp_hip_moments_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc != OK:
            raise HipError(rc, (lib().mcmcpp_hip_moments_last_error(self.h) or b"").decode())

    def add_steps(self, steps, slice_interval=1):
        """steps[(n, W, D)] (C-contiguous); every slice_interval-th step is used, starting with the first."""
        steps = np.ascontiguousarray(steps, dtype=np_dtype(self.dtype))
        assert steps.ndim == 3 and steps.shape[1:] == (self.W, self.D)
        used = (steps.shape[0] + slice_interval - 1) // slice_interval
        self._check(lib().mcmcpp_hip_moments_add_steps(self.h, _ptr(steps), used, slice_interval))

    def add_device_steps(self, device_ptr, n_steps):
        """n_steps contiguous stored steps in device memory (an integer address, e.g. torch.Tensor.data_ptr())."""
        self._check(lib().mcmcpp_hip_moments_add_device_steps(self.h, C.c_void_p(device_ptr), n_steps))

    def finish(self):
        t = np_dtype(self.dtype)
        mean, cov, corr = np.zeros(self.D, t), np.zeros((self.D, self.D), t), np.zeros((self.D, self.D), t)
        n = C.c_int64(0)
        self._check(lib().mcmcpp_hip_moments_finish(self.h, C.byref(n), _ptr(mean), _ptr(cov), _ptr(corr)))
        return n.value, mean, cov, corr

    def reset(self):
        self._check(lib().mcmcpp_hip_moments_reset(self.h))

    def close(self):
        if self.h:
            lib().mcmcpp_hip_moments_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def autocorr_times(steps, walkers_to_use=0, window_scaling=4, want_functions=False, dtype=None, device=-1):
    """Device-side Analysis::AutoCorrCalc::calcAutoCorrTimes (include/mcmcpp_hip.h, mcmcpp_hip_autocorr_times).

    steps: array [(n, W, D)] or a sequence of n arrays [(W, D)] (stored steps, oldest first).  Returns times[D], or
    (times, functions[(D, n)]) with want_functions."""
    if isinstance(steps, np.ndarray):
        if dtype is None:
            dtype = F32 if steps.dtype == np.float32 else F64
        steps = np.ascontiguousarray(steps, dtype=np_dtype(dtype))
        assert steps.ndim == 3
        blocks = [steps[i] for i in range(steps.shape[0])]
    else:
        if dtype is None:
            dtype = F32 if np.asarray(steps[0]).dtype == np.float32 else F64
        blocks = [np.ascontiguousarray(b, dtype=np_dtype(dtype)) for b in steps]
    n = len(blocks)
    W, D = blocks[0].shape
    assert all(b.shape == (W, D) for b in blocks)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in blocks])
    t = np_dtype(dtype)
    times = np.zeros(D, t)
    functions = np.zeros((D, n), t) if want_functions else None
    rc = lib().mcmcpp_hip_autocorr_times(dtype, device, ptrs, n, W, D, walkers_to_use, window_scaling, _ptr(times), _ptr(functions))
    if rc != OK:
        raise HipError(rc, (lib().mcmcpp_hip_autocorr_last_error() or b"").decode())
    return (times, functions) if want_functions else times


def autocorr_times_device(device_ptr, n_steps, W, D, dtype=F64, walkers_to_use=0, window_scaling=4, device=-1):
    """The same for n_steps contiguous stored steps in device memory (an integer address, e.g. torch.Tensor.data_ptr())."""
    times = np.zeros(D, np_dtype(dtype))
    rc = lib().mcmcpp_hip_autocorr_times_device(dtype, device, C.c_void_p(device_ptr), n_steps, W, D, walkers_to_use, window_scaling, _ptr(times), None)
    if rc != OK:
        raise HipError(rc, (lib().mcmcpp_hip_autocorr_last_error() or b"").decode())
    return times
