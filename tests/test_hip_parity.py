"""GPU parity tests: the HIP path (through the C ABI of libmcmcpp_hip.so) against the reference's golden
vectors and against the oracle on the same seeded inputs.

Bar: bit-exact positions, log-posteriors, per-step accepted counts and per-walker counters.  The only
quantities that may differ in the last ulp between device and host are the two `log` calls feeding the
accept comparison (StretchMove.h:110,113; OCML vs glibc); they are never stored, and a decision they
could flip is counted by both sides as a `near_tie` -- every case below asserts there were none, so
equality is exact, not approximate."""
import json
import os

import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po
from tests.goldens import DIGEST, GOLDEN_DIR, SMALL, Golden

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["full_step", "half_step"])
def step_path(request, monkeypatch):
    """run() steps small ensembles with one launch per ensemble step (full_step_kernel.hpp) and large ones with
    one launch per half-step; the cases that take this fixture are checked on both paths."""
    monkeypatch.setenv("MCMCPP_HIP_FULL_STEP", "1" if request.param == "full_step" else "0")
    return request.param


def _params_for(calc, D, t, rng):
    if calc == po.CALC_DENSE_GAUSSIAN:
        a = rng.standard_normal((D, D))
        return (a @ a.T / D + np.eye(D)).astype(t).ravel()
    if calc == po.CALC_ROSENBROCK:
        return np.array([1.0, 100.0, 0.05], dtype=t)
    if calc == po.CALC_SKEWED_GAUSSIAN_2D:
        return np.array([0.13], dtype=t)
    return None


@pytest.mark.parametrize("dtype", [po.F64, po.F32])
@pytest.mark.parametrize("calc,D", [(po.CALC_ISO_GAUSSIAN, 1), (po.CALC_ISO_GAUSSIAN, 2), (po.CALC_ISO_GAUSSIAN, 7),
                                    (po.CALC_ISO_GAUSSIAN, 32), (po.CALC_ISO_GAUSSIAN, 100),
                                    (po.CALC_ISO_GAUSSIAN, 513), (po.CALC_ISO_GAUSSIAN, 1024),
                                    (po.CALC_DENSE_GAUSSIAN, 3), (po.CALC_DENSE_GAUSSIAN, 32),
                                    (po.CALC_DENSE_GAUSSIAN, 64), (po.CALC_DENSE_GAUSSIAN, 130),
                                    (po.CALC_ROSENBROCK, 2), (po.CALC_ROSENBROCK, 5), (po.CALC_ROSENBROCK, 32),
                                    (po.CALC_ROSENBROCK, 33), (po.CALC_ROSENBROCK, 300),
                                    (po.CALC_SKEWED_GAUSSIAN_2D, 2)])
def test_device_calculators_bit_exact(calc, D, dtype):
    t = po.np_dtype(dtype)
    rng = np.random.default_rng(D * 7 + calc)
    W = 2 * D + 2 + (2 * D) % 2
    params = _params_for(calc, D, t, rng)
    pos = (rng.standard_normal((301, D)) * 1.5).astype(t)
    want = po.Oracle(W, D, calc, params, dtype=dtype).logp(pos)
    got = capi.HipSampler(W, D, calc, params, dtype=dtype).calc_logp(pos)
    np.testing.assert_array_equal(got, want)


def _run_hip_against_golden(g, **kw):
    s = capi.HipSampler(g.W, g.D, g.calc, g.params, seed=g.seed, dtype=g.dtype, alpha=g.alpha, **kw)
    s.set_state(g.init_pos, g.init_logp)
    done, acc_calls = 0, []
    for k in sorted(set(g.checked_steps + [g.steps])):
        chain, acc = s.run(k - done, interval=g.slicing)
        acc_calls.append(acc.reshape(k - done, g.slicing).sum(axis=1))
        done = k
        if k in g.checked_steps:
            g.check_chain_step(k, chain[-1])
    np.testing.assert_array_equal(np.concatenate(acc_calls), g.accepted_per_call)
    c = s.counters()
    assert c["accepted"] + g.W == g.accepted_total          # reference counts the initial placement
    assert g.W * (1 + c["ensemble_steps"]) == g.total_steps
    assert c["near_ties"] == 0 and c["redraws"] == 0
    return s


@pytest.mark.parametrize("name", SMALL + DIGEST)
def test_hip_matches_reference_golden(name, step_path):
    _run_hip_against_golden(Golden(name))


@pytest.mark.parametrize("name", ["iso100x7", "dense96x16", "skewed320x2"])
def test_hip_matches_reference_golden_without_graphs(name):
    _run_hip_against_golden(Golden(name), graph_steps=-1)


def _oracle_and_hip(W, D, calc, dtype, seed, steps, interval=1, salt=3, **kw):
    t = po.np_dtype(dtype)
    rng = np.random.default_rng(W + D)
    params = _params_for(calc, D, t, rng)
    orc = po.Oracle(W, D, calc, params, seed=seed, dtype=dtype)
    pos = po.init_positions(dtype, W, D, salt=salt)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    hip = capi.HipSampler(W, D, calc, params, seed=seed, dtype=dtype, **kw)
    hip.set_state(pos, logp)
    return orc, hip


def _assert_same_state(orc, hip, ties_expected=False):
    """ties_expected: fp32 runs of a few hundred thousand updates meet decisions inside the guard band (6e-7 of the
    magnitudes involved -- about one update in 10^5); the chains are compared bit for bit all the same, and device and
    oracle must have flagged equally many."""
    for a, b, what in zip(hip.get_state(), orc.get_state(), ("positions", "logp", "n_accept")):
        np.testing.assert_array_equal(a, b, err_msg=what)
    c = hip.counters()
    if ties_expected:
        assert c["near_ties"] == orc.near_ties and c["redraws"] == orc.redraws == 0
    else:
        assert c["near_ties"] == 0 == orc.near_ties and c["redraws"] == orc.redraws == 0


@pytest.mark.parametrize("W,D,calc,dtype,steps,interval", [
    (34, 3, po.CALC_ISO_GAUSSIAN, po.F64, 200, 1),       # smallest legal ensembles, odd D (scalar row access)
    (6, 2, po.CALC_ISO_GAUSSIAN, po.F64, 500, 1),        # n = 3: non power of two partner bound
    (4, 1, po.CALC_ISO_GAUSSIAN, po.F32, 500, 3),        # D = 1: (D-1) ln z == 0
    (1026, 512, po.CALC_ISO_GAUSSIAN, po.F64, 12, 1),    # EPL > 2 path (D > 128)
    (2100, 1024, po.CALC_ISO_GAUSSIAN, po.F32, 6, 1),    # maximum D
    (300, 64, po.CALC_DENSE_GAUSSIAN, po.F64, 40, 2),
    (150, 64, po.CALC_DENSE_GAUSSIAN, po.F32, 40, 1),
    (700, 130, po.CALC_DENSE_GAUSSIAN, po.F64, 8, 1),
    (4098, 32, po.CALC_ROSENBROCK, po.F32, 30, 1),       # ragged last wavefront
    (1000, 9, po.CALC_ROSENBROCK, po.F64, 60, 4),
])
def test_hip_equals_oracle_fresh_cases(W, D, calc, dtype, steps, interval, step_path):
    orc, hip = _oracle_and_hip(W, D, calc, dtype, seed=12345, steps=steps, interval=interval)
    oc, oa = orc.run(steps, interval=interval, mode=po.MODE_COUNTER, threads=4)
    hc, ha = hip.run(steps, interval=interval)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip)


def test_two_level_jump_table_with_non_power_of_two_half(monkeypatch, step_path):
    monkeypatch.setenv("MCMCPP_HIP_TASK_TABLE_MB", "0")
    orc, hip = _oracle_and_hip(1000, 9, po.CALC_ROSENBROCK, po.F64, seed=5, steps=0)
    oc, oa = orc.run(40)
    hc, ha = hip.run(40)
    np.testing.assert_array_equal(hc, oc)
    np.testing.assert_array_equal(ha, oa)
    _assert_same_state(orc, hip)


def test_seed_and_stream_are_honoured():
    for seed, stream in [(-7 & (2**64 - 1), 0), (2**40 + 1, 5)]:
        t = np.float64
        orc = po.Oracle(128, 8, po.CALC_ISO_GAUSSIAN, None, seed=seed, stream=stream)
        pos = po.init_positions(po.F64, 128, 8, salt=1)
        logp = orc.logp(pos)
        orc.set_state(pos, logp)
        hip = capi.HipSampler(128, 8, po.CALC_ISO_GAUSSIAN, None, seed=seed, stream=stream)
        hip.set_state(pos, logp)
        oc, oa = orc.run(50)
        hc, ha = hip.run(50)
        np.testing.assert_array_equal(hc, oc)
        np.testing.assert_array_equal(ha, oa)


def test_resume_reset_and_unsaved_runs_are_seamless(step_path):
    orc, hip = _oracle_and_hip(512, 16, po.CALC_ROSENBROCK, po.F64, seed=3, steps=0)
    oc, oa = orc.run(90)
    c1, a1 = hip.run(20)
    _, a2 = hip.run(5, interval=2, save_chain=False)           # 10 unsaved-to-host steps
    hip.reset_counters()                                       # EnsembleSampler::reset keeps positions/stream
    assert hip.counters()["accepted"] == 0 and hip.counters()["ensemble_steps"] == 0
    c3, a3 = hip.run(60, want_accepted=False)
    np.testing.assert_array_equal(c1, oc[:20])
    np.testing.assert_array_equal(c3, oc[30:])
    np.testing.assert_array_equal(np.concatenate([a1, a2]), oa[:30])
    assert a3 is None
    pos, logp, nacc = hip.get_state()
    opos, ologp, onacc = orc.get_state()
    np.testing.assert_array_equal(pos, opos)
    np.testing.assert_array_equal(logp, ologp)
    assert int(nacc.sum()) == int(oa[30:].sum())               # counters restarted at the reset


@pytest.mark.parametrize("dtype", [po.F64, po.F32])
@pytest.mark.parametrize("W,D", [(4096 + 6, 32), (600, 18), (2048, 26)])
def test_matrix_core_kernel_is_bit_exact(monkeypatch, W, D, dtype, step_path):
    """The MFMA variants of the step kernels against the oracle, including ragged last wavefronts and padded
    dimensions (17 <= D <= 32, even): v_mfma_f64_16x16x4_f64 and, since round 3, v_mfma_f32_16x16x4_f32 (both are the
    host calculator's fma chain in ascending k: tools/mfma_probe.hip, tools/mfma_f32_probe.hip)."""
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", "0")
    orc, hip = _oracle_and_hip(W, D, po.CALC_DENSE_GAUSSIAN, dtype, seed=77, steps=0)
    oc, oa = orc.run(30, interval=2, mode=po.MODE_COUNTER, threads=4)
    hc, ha = hip.run(30, interval=2)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip, ties_expected=dtype == po.F32)


def test_matrix_core_kernels_with_sixteen_walkers_per_wavefront_fp32(monkeypatch):
    """The 16-walkers-per-wavefront matrix-core half-step kernel in fp32 (rows 4g + q of the tile: every register of a lane
    group in use), with a ragged last wavefront, and against the plain fp32 kernels (MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS=-1)."""
    monkeypatch.setenv("MCMCPP_HIP_FULL_STEP", "0")
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS", "1")
    orc, hip = _oracle_and_hip(8192 + 10, 32, po.CALC_DENSE_GAUSSIAN, po.F32, seed=3, steps=0)
    oc, oa = orc.run(12, interval=2, mode=po.MODE_COUNTER, threads=4)
    hc, ha = hip.run(12, interval=2)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip, ties_expected=True)
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", "-1")
    orc2, plain = _oracle_and_hip(8192 + 10, 32, po.CALC_DENSE_GAUSSIAN, po.F32, seed=3, steps=0)
    pc, pa = plain.run(12, interval=2)
    np.testing.assert_array_equal(pc, hc)
    np.testing.assert_array_equal(pa, ha)


@pytest.mark.parametrize("late", ["0", "-1"])
@pytest.mark.parametrize("dtype", [po.F64, po.F32])
@pytest.mark.parametrize("W,D,chains", [(8192 + 74, 32, 1), (2048 + 6, 26, 1), (1024 + 38, 32, 3)])
def test_matrix_core_kernel_with_late_draws_four_wavefronts_per_simd(monkeypatch, W, D, chains, dtype, late):
    """The 16-walker matrix-core half-step kernel in the form large launches get (MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS): next draws
    behind the accept, 118 registers, four wavefronts per SIMD; P^T shared by the workgroup through LDS behind a barrier that
    wavefronts without walkers reach too, the accepted counters by adds that return nothing, stored steps re-read from the
    rows, the workgroup's next draws divided by kind among its wavefronts (those without walkers make their share).  Ragged
    last wavefronts and workgroups, padded dimensions, several chains per launch, stored steps at an interval.  late = "-1":
    the same kernel with the draws in the gather's shadow (what launches of 18 432 .. 49 151 updates take)."""
    monkeypatch.setenv("MCMCPP_HIP_FULL_STEP", "0")
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", "0")
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS", "1")
    monkeypatch.setenv("MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS", late)
    if chains == 1:
        orc, hip = _oracle_and_hip(W, D, po.CALC_DENSE_GAUSSIAN, dtype, seed=11, steps=0)
        oc, oa = orc.run(14, interval=2, mode=po.MODE_COUNTER, threads=4)
        hc, ha = hip.run(14, interval=2)
        np.testing.assert_array_equal(ha, oa)
        np.testing.assert_array_equal(hc, oc)
        _assert_same_state(orc, hip, ties_expected=dtype == po.F32)
        return
    t = po.np_dtype(dtype)
    params = _params_for(po.CALC_DENSE_GAUSSIAN, D, t, np.random.default_rng(W + D))
    orcs = []
    for k in range(chains):
        orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, params, seed=11 + k, dtype=dtype)
        pos = po.init_positions(dtype, W, D, salt=20 + k)
        logp = orc.logp(pos)
        orc.set_state(pos, logp)
        orcs.append((orc, pos, logp))
    hip = capi.HipSampler(W, D, po.CALC_DENSE_GAUSSIAN, params, seed=11, dtype=dtype, num_chains=chains)
    hip.set_state(np.stack([o[1] for o in orcs]), np.stack([o[2] for o in orcs]))
    hc, ha = hip.run(10, interval=2)
    for k, (orc, _, _) in enumerate(orcs):
        oc, oa = orc.run(10, interval=2, mode=po.MODE_COUNTER, threads=4)
        np.testing.assert_array_equal(ha[k], oa, err_msg="chain %d" % k)
        np.testing.assert_array_equal(hc[k], oc, err_msg="chain %d" % k)
    pos, logp, nacc = hip.get_state()
    for k, (orc, _, _) in enumerate(orcs):
        opos, ologp, onacc = orc.get_state()
        np.testing.assert_array_equal(pos[k], opos)
        np.testing.assert_array_equal(logp[k], ologp)
        np.testing.assert_array_equal(nacc[k], onacc)


@pytest.mark.parametrize("batch", ["0", "1", "7", "128"])
def test_draw_records_made_ahead_in_batches_of_any_length(monkeypatch, batch):
    """The matrix-core full-step launches of one whole ensemble read draw records made ahead by a launch of their own,
    MCMCPP_HIP_BATCH_DRAWS ensemble steps at a time (0: their own extra wavefronts make them, as for chains and split
    ensembles): the chain must not depend on it, across runs that end inside a batch, with and without stored steps."""
    monkeypatch.setenv("MCMCPP_HIP_BATCH_DRAWS", batch)
    orc, hip = _oracle_and_hip(2048 + 6, 32, po.CALC_DENSE_GAUSSIAN, po.F64, seed=5, steps=0)
    oc, oa = orc.run(45, interval=3, mode=po.MODE_COUNTER, threads=4)
    got = [hip.run(k, interval=3) for k in (1, 9, 5)]
    hip.run(10, interval=3, save_chain=False)
    got.append(hip.run(20, interval=3))
    np.testing.assert_array_equal(np.concatenate([g[0] for g in got[:3]]), oc[:15])
    np.testing.assert_array_equal(got[3][0], oc[25:])
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got[:3]]), oa[:45])
    _assert_same_state(orc, hip)


@pytest.mark.parametrize("pieces", ["1", "3", "8"])
def test_draw_records_of_the_next_replay_made_on_a_branch_of_this_one(monkeypatch, pieces):
    """MCMCPP_HIP_FILL_BRANCH = p: a whole graph replay makes the NEXT replay's draw records beside its own step launches, in
    p pieces on a parallel branch of the graph (piece k forked off behind the last step that read its record sets).  Whole
    replays, remainders that use records made ahead, remainders that make their own, runs that store steps and runs that do
    not, a seek in between: the chain and the accepted counts are the oracle's."""
    monkeypatch.setenv("MCMCPP_HIP_FILL_BRANCH", pieces)
    monkeypatch.setenv("MCMCPP_HIP_GRAPH_STEPS", "8")
    orc, hip = _oracle_and_hip(2048 + 6, 32, po.CALC_DENSE_GAUSSIAN, po.F64, seed=5, steps=0)
    oc, oa = orc.run(104, interval=1, mode=po.MODE_COUNTER, threads=4)
    at = 0
    for steps, save in ((3, False), (27, False), (16, True), (8, False), (9, False), (17, True), (24, False)):
        hc, ha = hip.run(steps, interval=1, save_chain=save)
        np.testing.assert_array_equal(ha, oa[at:at + steps])
        if save:
            np.testing.assert_array_equal(hc, oc[at:at + steps])
        at += steps
    assert at == 104
    _assert_same_state(orc, hip)
    # back to step 40 with the walkers as they stand: records made ahead for another step must not be used
    hip.seek(40)
    orc.seek(40)
    oc2, oa2 = orc.run(20, interval=1, mode=po.MODE_COUNTER, threads=4)
    hc2, ha2 = hip.run(20, interval=1)
    np.testing.assert_array_equal(hc2, oc2)
    np.testing.assert_array_equal(ha2, oa2)
    _assert_same_state(orc, hip)


def test_checkpoint_and_resume_in_a_new_handle(step_path):
    """get_state + the number of steps done is a complete checkpoint: a fresh handle resumes the trajectory."""
    orc, hip = _oracle_and_hip(1024, 16, po.CALC_DENSE_GAUSSIAN, po.F64, seed=21, steps=0)
    oc, oa = orc.run(70)
    hip.run(25, save_chain=False, want_accepted=False)
    pos, logp, _ = hip.get_state()
    rng = np.random.default_rng(1024 + 16)
    params = _params_for(po.CALC_DENSE_GAUSSIAN, 16, np.float64, rng)
    fresh = capi.HipSampler(1024, 16, po.CALC_DENSE_GAUSSIAN, params, seed=21)
    fresh.set_state(pos, logp)
    fresh.seek(25)
    chain, acc = fresh.run(45)
    np.testing.assert_array_equal(chain, oc[25:])
    np.testing.assert_array_equal(acc, oa[25:])


def test_wave_mapping_and_chunking_do_not_change_results(monkeypatch):
    base = None
    for env in [{}, {"MCMCPP_HIP_PASSES": "1"}, {"MCMCPP_HIP_PASSES": "4"}, {"MCMCPP_HIP_PASSES": "16"},
                {"MCMCPP_HIP_CHAIN_SUBCHUNK_MB": "1", "MCMCPP_HIP_GRAPH_STEPS": "7"},
                {"MCMCPP_HIP_TASK_TABLE_MB": "0"},                         # two-level jump table (very large ensembles)
                {"MCMCPP_HIP_TASK_TABLE_MB": "0", "MCMCPP_HIP_PASSES": "8"},
                {"MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS": "0"},              # matrix-core kernel
                {"MCMCPP_HIP_FULL_STEP": "0"},                             # one launch per half-step
                {"MCMCPP_HIP_FULL_STEP": "0", "MCMCPP_HIP_NO_DRAW_WAVE": "1"},
                {"MCMCPP_HIP_NO_DRAW_WAVE": "1"},                          # the updating wavefronts make the next draws
                {"MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS": "-1"},              # generic kernels for the dense target
                {"MCMCPP_HIP_COPY_STREAM": "1", "MCMCPP_HIP_CHAIN_SUBCHUNK_MB": "1", "MCMCPP_HIP_TRICKLE": "0"},
                {"MCMCPP_HIP_TRICKLE": "0"},                               # stored steps by device-to-host copies
                {"MCMCPP_HIP_CHAIN_SUBCHUNK_MB": "1"}]:                    # the smallest ring of stored steps
        for k in ("MCMCPP_HIP_PASSES", "MCMCPP_HIP_CHAIN_SUBCHUNK_MB", "MCMCPP_HIP_GRAPH_STEPS", "MCMCPP_HIP_TASK_TABLE_MB",
                  "MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", "MCMCPP_HIP_FULL_STEP", "MCMCPP_HIP_NO_DRAW_WAVE", "MCMCPP_HIP_COPY_STREAM",
                  "MCMCPP_HIP_TRICKLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        orc, hip = _oracle_and_hip(2048 + 2, 32, po.CALC_DENSE_GAUSSIAN, po.F64, seed=9, steps=0)
        chain, acc = hip.run(25)
        state = hip.get_state()
        if base is None:
            oc, oa = orc.run(25, mode=po.MODE_COUNTER, threads=4)
            np.testing.assert_array_equal(chain, oc)
            np.testing.assert_array_equal(acc, oa)
            base = (chain, acc, state)
        else:
            np.testing.assert_array_equal(chain, base[0])
            np.testing.assert_array_equal(acc, base[1])
            for a, b in zip(state, base[2]):
                np.testing.assert_array_equal(a, b)


def test_full_step_runs_interleave_with_half_step_calls():
    """run() (one launch per ensemble step, ping-pong position buffers) leaves everything where the half-step entry
    points expect it: odd step counts, single steps, half_step_async in between, get_state after each."""
    W, D = 640, 12
    orc, hip = _oracle_and_hip(W, D, po.CALC_ROSENBROCK, po.F64, seed=31, steps=0)
    oc, oa = orc.run(1 + 3 + 2 + 1 + 5)
    chunks = []
    c, _ = hip.run(1)
    chunks.append(c)
    np.testing.assert_array_equal(hip.get_state()[0], oc[0])
    c, _ = hip.run(3)
    chunks.append(c)
    for _ in range(2):                      # two ensemble steps through the sharded driver's entry point
        hip.half_step_async(0)
        hip.half_step_async(1)
    hip.synchronize()
    np.testing.assert_array_equal(hip.get_state()[0], oc[5])
    c, _ = hip.run(1)
    chunks.append(c)
    c, a = hip.run(5)
    chunks.append(c)
    np.testing.assert_array_equal(np.concatenate(chunks), np.concatenate([oc[:4], oc[6:]]))
    np.testing.assert_array_equal(a, oa[7:])
    _assert_same_state(orc, hip)


@pytest.mark.parametrize("W,D,dtype,interval,n_saved", [
    (2048, 32, po.F64, 1, 40),      # every step stored: each launch forwards a whole step
    (2048, 32, po.F64, 3, 11),
    (1000, 9, po.F64, 7, 5),        # 72 000-byte steps, slices that do not divide them
    (518, 5, po.F32, 2, 9),         # 10 360-byte steps (a multiple of 8, not of 16): the copy-engine path
    (512, 16, po.F32, 50, 1),       # a single stored step: nothing to forward
    (4096, 32, po.F64, 100, 3),
])
def test_stored_steps_forwarded_by_the_launches(W, D, dtype, interval, n_saved):
    """The chain path of the full-step kernels (trickle_stored_step): every stored step arrives intact whatever the
    slicing interval, ring size and run length."""
    orc, hip = _oracle_and_hip(W, D, po.CALC_ROSENBROCK, dtype, seed=8, steps=0)
    for part in (n_saved, 2):       # a second, short run on the same handle
        oc, oa = orc.run(part, interval=interval, mode=po.MODE_COUNTER, threads=4)
        hc, ha = hip.run(part, interval=interval)
        np.testing.assert_array_equal(hc, oc)
        np.testing.assert_array_equal(ha, oa)
    _assert_same_state(orc, hip)


def test_hip_reproduces_reference_own_test_run():
    """The reference's SkewedGaussian/StretchMove test end to end: 320 x 2, slicing 30, 40 019 stored steps
    (1.2 M ensemble steps); accepted/total exactly as the reference prints them."""
    want = json.load(open(os.path.join(GOLDEN_DIR, "reference_skewed_test.json")))
    g = Golden("skewed320x2")
    s = capi.HipSampler(g.W, g.D, g.calc, g.params, seed=0)
    s.set_state(g.init_pos, g.init_logp)
    chain, _ = s.run(want["stored_steps"], interval=want["slicing"], want_accepted=False)
    c = s.counters()
    assert c["near_ties"] == 0
    assert c["accepted"] + g.W == want["accepted_total"]
    assert g.W * (1 + c["ensemble_steps"]) == want["total_steps"]
    # sliceAndBurnChain(1, 20) then covariance, as the reference's main does (analytic: 1.13, 0.435, 0.2825)
    x = chain[20:].reshape(-1, 2)
    cov = np.cov(x.T)
    assert abs(cov[0, 0] - 1.13) < 0.02 and abs(cov[0, 1] - 0.435) < 0.01 and abs(cov[1, 1] - 0.2825) < 0.005


# ---- BASELINE.json full sizes ---------------------------------------------------------------------------

def test_config3_full_size_against_oracle():
    """65 536 walkers x 32-dim Rosenbrock (BASELINE config 3): 6 steps against the multi-threaded oracle."""
    orc, hip = _oracle_and_hip(65536, 32, po.CALC_ROSENBROCK, po.F64, seed=0, steps=0, salt=0)
    oc, oa = orc.run(6, mode=po.MODE_COUNTER, threads=8)
    hc, ha = hip.run(6)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip)


def test_config5_shape_single_gpu_against_oracle():
    """131 072 walkers x 64-dim Gaussian (BASELINE config 5's ensemble) on one GPU: 3 steps."""
    orc, hip = _oracle_and_hip(131072, 64, po.CALC_ISO_GAUSSIAN, po.F64, seed=0, steps=0, salt=0)
    oc, oa = orc.run(3, mode=po.MODE_COUNTER, threads=8)
    hc, ha = hip.run(3)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip)


def test_million_walker_ensemble_and_empty_run():
    """Maximum sizes: 2^20 + 2 walkers (two-level jump table, ragged last wavefront) for two steps against the
    multi-threaded oracle; and the empty request (0 stored steps) is a no-op."""
    orc, hip = _oracle_and_hip(2**20 + 2, 8, po.CALC_ISO_GAUSSIAN, po.F64, seed=4, steps=0, salt=0)
    chain, acc = hip.run(0)
    assert chain.shape[0] == 0 and acc.size == 0 and hip.counters()["ensemble_steps"] == 0
    oc, oa = orc.run(2, mode=po.MODE_COUNTER, threads=8)
    hc, ha = hip.run(2)
    np.testing.assert_array_equal(ha, oa)
    np.testing.assert_array_equal(hc, oc)
    _assert_same_state(orc, hip)


def test_config2_statistical_properties_long_run():
    """16 384 x 32 correlated Gaussian (BASELINE config 2), 600 steps: size-independent checks --
    acceptance in the stretch-move range for D = 32, and the sample covariance of the last stored steps
    approaches Sigma_ij = 0.5^|i-j| (the target's covariance)."""
    D, W = 32, 16384
    P = np.zeros((D, D))
    rho = 0.5
    d = 1 - rho * rho
    for i in range(D):
        P[i, i] = (1.0 if i in (0, D - 1) else 1 + rho * rho) / d
        if i + 1 < D:
            P[i, i + 1] = P[i + 1, i] = -rho / d
    hip = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=1)
    pos = po.init_positions(po.F64, W, D, salt=0)
    hip.set_state(pos, hip.calc_logp(pos))
    hip.run(1, interval=500, save_chain=False)
    chain, acc = hip.run(2, interval=50)
    rate = acc.mean() / W
    assert 0.2 < rate < 0.45, rate
    cov = np.cov(chain.reshape(-1, D).T)
    sigma = rho ** np.abs(np.subtract.outer(np.arange(D), np.arange(D)))
    assert np.abs(cov - sigma).max() < 0.06, np.abs(cov - sigma).max()
