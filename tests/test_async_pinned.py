"""GPU tests of the two host-boundary additions of the chain store:

* stored steps forwarded by the step launches straight into PINNED host memory handed out by the library
  (mcmcpp_hip_host_alloc: the facade's Chain blocks) -- no staging ring, no host copy;
* mcmcpp_hip_run_async / wait_stored / run_wait: the run on a worker thread, stored steps announced as they become
  complete in the caller's memory (what a PostStepAction looks at while the device keeps stepping,
  /root/reference/MCMCpp/EnsembleSampler.h:356-359).
Both must leave the chain bit-identical to the oracle's."""
import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _pair(W, D, calc, seed, params=None):
    orc = po.Oracle(W, D, calc, params, seed=seed)
    pos = po.init_positions(po.F64, W, D, salt=seed)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    hip = capi.HipSampler(W, D, calc, params, seed=seed)
    hip.set_state(pos, logp)
    return orc, hip


@pytest.mark.parametrize("W,D,n_saved,interval", [(4096, 32, 7, 1), (4096, 32, 5, 3), (2048, 16, 40, 1), (16384, 32, 4, 100),
                                                  (1024, 8, 1, 1), (1024, 8, 3, 700)])
def test_stored_steps_forwarded_into_pinned_chain_memory(W, D, n_saved, interval):
    orc, hip = _pair(W, D, po.CALC_ISO_GAUSSIAN, 3)
    block = capi.pinned_empty((n_saved + 2, W, D))
    block[:] = -7.0
    for first, count in ((0, n_saved), (n_saved, 2)):  # a second run appends to the same block
        want, want_acc = orc.run(count, interval=interval, mode=po.MODE_COUNTER, threads=4)
        chain, acc = hip.run(count, interval=interval, out=block[first:first + count])
        np.testing.assert_array_equal(acc, want_acc)
        np.testing.assert_array_equal(block[first:first + count], want)
    for got, want in zip(hip.get_state(), orc.get_state()):
        np.testing.assert_array_equal(got, want)
    assert hip.counters()["near_ties"] == 0


def test_pinned_and_pageable_destinations_agree_on_the_half_step_path(monkeypatch):
    """Ensembles stepped by the half-step kernels download stored steps by copies; a pinned destination changes nothing."""
    monkeypatch.setenv("MCMCPP_HIP_FULL_STEP", "0")
    orc, hip = _pair(4096, 32, po.CALC_ROSENBROCK, 5, np.array([1.0, 100.0, 0.05]))
    want, _ = orc.run(6, interval=2)
    block = capi.pinned_empty((6, 4096, 32))
    hip.run(6, interval=2, out=block)
    np.testing.assert_array_equal(block, want)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("W,D,n_saved,interval", [(4096, 32, 24, 1), (4096, 32, 6, 50), (16384, 32, 3, 300)])
def test_async_run_announces_complete_stored_steps(W, D, n_saved, interval, pinned):
    orc, hip = _pair(W, D, po.CALC_ISO_GAUSSIAN, 1)
    want, want_acc = orc.run(n_saved, interval=interval, mode=po.MODE_COUNTER, threads=4)
    out = capi.pinned_empty((n_saved, W, D)) if pinned else np.empty((n_saved, W, D))
    out[:] = np.nan
    chain, acc = hip.run_async(n_saved, interval=interval, out=out, want_accepted=True)
    for k in range(n_saved):
        hip.wait_stored(k + 1)
        np.testing.assert_array_equal(chain[k].copy(), want[k], err_msg="stored step %d was announced before it was complete" % k)
    hip.run_wait()
    np.testing.assert_array_equal(acc, want_acc)
    with pytest.raises(capi.HipError):
        hip.run_wait()  # nothing in flight any more
    for got, wanted in zip(hip.get_state(), orc.get_state()):
        np.testing.assert_array_equal(got, wanted)


def test_async_run_reports_a_failed_run():
    W, D = 1024, 8
    hip = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None)
    out = np.empty((2, W, D))
    hip.run_async(2, out=out)  # no set_state yet: the run fails on the worker
    with pytest.raises(capi.HipError):
        hip.wait_stored(1)
    with pytest.raises(capi.HipError):
        hip.run_wait()


@pytest.mark.parametrize("mover", [capi.MOVER_STRETCH, capi.MOVER_DIFFERENTIAL_EVOLUTION])
def test_handle_belongs_to_its_worker_between_run_async_and_run_wait(mover):
    """Between run_async and run_wait only wait_stored is allowed: every other entry point is refused (E_STATE) instead
    of racing with the worker, and destroying the handle first lets the run end (the worker uses the arena, streams and
    graphs the destructor frees)."""
    W, D = 16384, 32
    pos = po.init_positions(po.F64, W, D, salt=2)
    hip = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=4, mover=mover)
    hip.set_state(pos, hip.calc_logp(pos))
    chain, _ = hip.run_async(4, interval=500)
    for call in (lambda: hip.run(1), lambda: hip.get_state(), lambda: hip.set_state(pos, np.zeros(W)), lambda: hip.reset_counters(),
                 lambda: hip.counters(), lambda: hip.seek(0), lambda: hip.calc_logp(pos), lambda: hip.synchronize(),
                 lambda: hip.run_async(1)):
        with pytest.raises(capi.HipError) as e:
            call()
        assert e.value.code == 5 and "asynchronous run" in str(e.value)
    hip.wait_stored(1)
    hip.run_wait()
    first = chain.copy()
    assert hip.counters()["ensemble_steps"] == 2000
    # the same run again from the same state; this time the handle is destroyed while the worker is stepping
    hip.set_state(pos, hip.calc_logp(pos))
    again, _ = hip.run_async(4, interval=500)
    hip.close()  # joins the worker, then frees
    np.testing.assert_array_equal(again, first)
