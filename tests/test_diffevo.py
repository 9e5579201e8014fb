"""Next row f3: Mover::DifferentialEvolution (reference MCMCpp/Movers/DifferentialEvolution.h:80-112).

CPU: the oracle's restatement (oracle/stretch_oracle_typed.inc: de_update_walker) against fixtures produced by the
reference's own mover inside its own EnsembleSampler (tests/golden/make_golden.py de_*): chain steps, accepted counts
per call and totals, bit for bit -- including half-ensemble sizes that are not powers of two (pcg's bounded_rand throws
draws away) and one so small that the second partner collides with the first every few updates (the redraw loop)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.goldens import DE_DIGEST, DE_SMALL, Golden


def run_oracle(g, split=None):
    orc = po.Oracle(g.W, g.D, g.calc, g.params, seed=g.seed, dtype=g.dtype, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    np.testing.assert_array_equal(orc.logp(g.init_pos), g.init_logp)
    orc.set_state(g.init_pos, g.init_logp)
    done, acc_calls = 0, []
    stops = sorted(set(g.checked_steps + [g.steps] + (split or [])))
    for k in stops:
        chain, acc = orc.run(k - done, interval=g.slicing, save_chain=True)
        acc_calls.append(acc.reshape(k - done, g.slicing).sum(axis=1))
        done = k
        if k in g.checked_steps:
            g.check_chain_step(k, chain[-1])
    np.testing.assert_array_equal(np.concatenate(acc_calls), g.accepted_per_call)
    pos, logp, nacc = orc.get_state()
    assert int(nacc.sum()) + g.W == g.accepted_total
    return orc


@pytest.mark.parametrize("name", DE_SMALL + DE_DIGEST)
def test_oracle_differential_evolution_matches_reference_golden(name):
    g = Golden(name)
    assert g.mover == po.MOVER_DIFFERENTIAL_EVOLUTION
    orc = run_oracle(g)
    n = g.W // 2
    if n & (n - 1) and n < 64:
        assert orc.redraws > 0  # the fixture does exercise the data-dependent draw count


def test_oracle_differential_evolution_resumes_across_calls():
    # the stream position of a later call includes every draw thrown away so far
    run_oracle(Golden("de_iso14x3"), split=[3, 77, 78, 200])


def test_oracle_differential_evolution_has_no_counter_mode():
    g = Golden("de_iso64x4")
    orc = po.Oracle(g.W, g.D, g.calc, g.params, seed=g.seed, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    orc.set_state(g.init_pos, g.init_logp)
    with pytest.raises(ValueError):
        orc.run(2, mode=po.MODE_COUNTER)


def test_oracle_reproduces_the_references_own_diffevo_test_run():
    """test/sequential/SkewedGaussian/DiffEvo/src/main.cpp (320 x 2 skewed Gaussian, slicing 10, 40 019 stored steps) end to
    end: accepted/total as the reference computes them (tests/golden/reference_skewed_diffevo_test.json)."""
    import json
    import os
    from tests.goldens import GOLDEN_DIR
    want = json.load(open(os.path.join(GOLDEN_DIR, "reference_skewed_diffevo_test.json")))
    g = Golden("skewed320x2")  # the test's initial placement
    orc = po.Oracle(g.W, g.D, g.calc, g.params, seed=0, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    orc.set_state(g.init_pos, g.init_logp)
    _, acc = orc.run(want["stored_steps"], interval=want["slicing"], save_chain=False)
    assert int(acc.sum()) + g.W == want["accepted_total"]
    assert g.W * (1 + want["stored_steps"] * want["slicing"]) == want["total_steps"]


@pytest.mark.skipif(not po.reference_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_differential_evolution_against_the_live_reference():
    W, D = 36, 5
    pos = po.init_positions(po.F64, W, D, salt=4)
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=9, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = orc.logp(pos)
    orc.set_state(pos, lp)
    chain, acc = orc.run(250)
    ref = po.reference_run(W, D, po.CALC_ISO_GAUSSIAN, None, 9, pos, lp, 1, 250, alpha_code=2)
    np.testing.assert_array_equal(ref["chain"][1:], chain)
    assert int(acc.sum()) + W == int(ref["accepted"][0])


# ---- the device path (mcmcpp_amd/csrc/diffevo.hip, diffevo_kernel.hpp) through the C ABI ---------------------------------
from mcmcpp_amd import capi  # noqa: E402


def run_device(g, split=None):
    s = capi.HipSampler(g.W, g.D, g.calc, g.params, seed=g.seed, dtype=g.dtype, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    np.testing.assert_array_equal(s.calc_logp(g.init_pos), g.init_logp)
    s.set_state(g.init_pos, g.init_logp)
    done, acc_calls = 0, []
    for k in sorted(set(g.checked_steps + [g.steps] + (split or []))):
        chain, acc = s.run(k - done, g.slicing)
        acc_calls.append(acc.reshape(k - done, g.slicing).sum(axis=1))
        done = k
        if k in g.checked_steps:
            g.check_chain_step(k, chain[-1])
    np.testing.assert_array_equal(np.concatenate(acc_calls), g.accepted_per_call)
    pos, logp, nacc = s.get_state()
    assert int(nacc.sum()) + g.W == g.accepted_total
    c = s.counters()
    assert c["near_ties"] == 0
    return s, c


@pytest.mark.gpu
@pytest.mark.parametrize("name", DE_SMALL + DE_DIGEST)
def test_device_differential_evolution_matches_reference_golden(name):
    g = Golden(name)
    s, c = run_device(g)
    orc = run_oracle(g)
    assert c["redraws"] == orc.redraws  # every draw thrown away was followed


@pytest.mark.gpu
def test_device_differential_evolution_resumes_across_calls():
    run_device(Golden("de_iso14x3"), split=[3, 77, 78, 200])


@pytest.mark.gpu
@pytest.mark.parametrize("W,D,calc,params,dt,steps,interval", [
    (4, 1, po.CALC_ISO_GAUSSIAN, None, po.F64, 2000, 1), (8, 1, po.CALC_ISO_GAUSSIAN, None, po.F64, 300, 1), (10, 2, po.CALC_ISO_GAUSSIAN, None, po.F64, 200, 3),
    (40, 3, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], po.F64, 150, 1), (70, 33, po.CALC_ISO_GAUSSIAN, None, po.F64, 60, 2),
    (140, 64, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], po.F64, 40, 1), (300, 130, po.CALC_ISO_GAUSSIAN, None, po.F64, 20, 1),
    (600, 257, po.CALC_ISO_GAUSSIAN, None, po.F64, 8, 1), (1026, 32, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], po.F64, 40, 1),
    (64, 7, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], po.F32, 200, 1), (4096, 16, po.CALC_ISO_GAUSSIAN, None, po.F32, 20, 5),
    (4100, 8, po.CALC_ISO_GAUSSIAN, None, po.F64, 30, 1),
    # BASELINE's larger ensembles (several planning batches each) and the widest walker the kernels are built for
    (65536, 32, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], po.F64, 70, 35), (131072, 64, po.CALC_ISO_GAUSSIAN, None, po.F64, 3, 1),
    (2200, 1024, po.CALC_ISO_GAUSSIAN, None, po.F64, 6, 1),
    # the dense Gaussian on the matrix cores in both element types: 8 walkers per wavefront, a ragged last wavefront, and
    # 16 per wavefront (from 32 768 walkers per half)
    (1030, 32, po.CALC_DENSE_GAUSSIAN, "ar1", po.F64, 30, 1), (1030, 26, po.CALC_DENSE_GAUSSIAN, "ar1", po.F32, 30, 1),
    (65536 + 20, 32, po.CALC_DENSE_GAUSSIAN, "ar1", po.F32, 4, 1), (65536 + 20, 32, po.CALC_DENSE_GAUSSIAN, "ar1", po.F64, 4, 1),
])
def test_device_differential_evolution_matches_the_oracle(W, D, calc, params, dt, steps, interval):
    if isinstance(params, str):
        from mcmcpp_amd import workloads
        params = workloads.ar1_precision(D, 0.5).ravel()
    pos = po.init_positions(dt, W, D, salt=6)
    orc = po.Oracle(W, D, calc, params, seed=21, dtype=dt, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = orc.logp(pos)
    orc.set_state(pos, lp)
    want_chain, want_acc = orc.run(steps, interval)
    s = capi.HipSampler(W, D, calc, params, seed=21, dtype=dt, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    s.set_state(pos, lp)
    chain, acc = s.run(steps, interval)
    np.testing.assert_array_equal(chain, want_chain)
    np.testing.assert_array_equal(acc, want_acc)
    for got, want in zip(s.get_state(), orc.get_state()):
        np.testing.assert_array_equal(got, want)
    c = s.counters()
    assert c["redraws"] == orc.redraws and c["ensemble_steps"] == steps * interval
    if dt == po.F64:  # (fp32 decisions within a few ulp of flipping do occur at this count; chain equality above is the test)
        assert c["near_ties"] == 0 and orc.near_ties == 0


@pytest.mark.gpu
@pytest.mark.parametrize("batch,scan_run", [(1, 8), (3, 1), (7, 5), (64, 32)])
def test_device_differential_evolution_planning_batches_of_any_length(monkeypatch, batch, scan_run):
    # the planner's batch length and scan run are free parameters of the device path: the chain must not depend on them
    monkeypatch.setenv("MCMCPP_HIP_DE_BATCH", str(batch))
    monkeypatch.setenv("MCMCPP_HIP_DE_SCAN_RUN", str(scan_run))
    W, D, steps = 22, 2, 300
    pos = po.init_positions(po.F64, W, D, salt=3)
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=5, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = orc.logp(pos)
    orc.set_state(pos, lp)
    want_chain, want_acc = orc.run(steps, 1)
    s = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=5, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    s.set_state(pos, lp)
    got = [s.run(k, 1) for k in (1, 2, 97, 200)]  # (runs that begin and end anywhere inside a batch)
    np.testing.assert_array_equal(np.concatenate([g[0] for g in got]), want_chain)
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got]), want_acc)
    assert s.counters()["redraws"] == orc.redraws


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(graph_steps=3), dict(graph_steps=-1), dict(hip_stream=0)])
def test_device_differential_evolution_short_replays_and_plain_launches(kw):
    # graphs of three ensemble steps (every phase of a planning batch begins one), plain launches, and the legacy default
    # stream (on which HIP cannot capture: plain launches too)
    W, D, steps = 300, 5, 170
    pos = po.init_positions(po.F64, W, D, salt=1)
    orc = po.Oracle(W, D, po.CALC_ROSENBROCK, [1.0, 100.0, 0.05], seed=3, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = orc.logp(pos)
    orc.set_state(pos, lp)
    want_chain, want_acc = orc.run(steps, 1)
    s = capi.HipSampler(W, D, capi.CALC_ROSENBROCK, [1.0, 100.0, 0.05], seed=3, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION, **kw)
    s.set_state(pos, lp)
    chain, acc = s.run(steps, 1)
    np.testing.assert_array_equal(chain, want_chain)
    np.testing.assert_array_equal(acc, want_acc)
    assert s.counters()["redraws"] == orc.redraws


@pytest.mark.gpu
def test_device_differential_evolution_dense_c2_shape_statistics():
    # BASELINE's C2 target under the other mover: same stationary distribution (variance 1 per parameter for the AR(1) covariance)
    from tests.golden.make_golden import ar1_precision
    W, D = 4096, 32
    P = ar1_precision(D, 0.5, np.float64)
    pos = po.init_positions(po.F64, W, D, salt=0)
    s = capi.HipSampler(W, D, capi.CALC_DENSE_GAUSSIAN, P.ravel(), seed=1, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    s.set_state(pos, s.calc_logp(pos))
    s.run(1, 600, save_chain=False)
    chain, acc = s.run(10, 20)
    var = chain.reshape(-1, D).var(axis=0)
    assert np.all(np.abs(var - 1.0) < 0.08), var
    rate = acc.sum() / (W * acc.size)
    assert 0.2 < rate < 0.6, rate


@pytest.mark.gpu
def test_device_differential_evolution_long_intervals_in_pieces():
    # 5 stored steps 30 000 ensemble steps apart: the accepted counters leave the device in several pieces
    W, D, n_saved, interval = 14, 3, 5, 30000
    pos = po.init_positions(po.F64, W, D, salt=2)
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=4, mover=po.MOVER_DIFFERENTIAL_EVOLUTION)
    lp = orc.logp(pos)
    orc.set_state(pos, lp)
    want_chain, want_acc = orc.run(n_saved, interval)
    s = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=4, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    s.set_state(pos, lp)
    chain, acc = s.run(n_saved, interval)
    np.testing.assert_array_equal(chain, want_chain)
    np.testing.assert_array_equal(acc, want_acc)
    assert s.counters()["redraws"] == orc.redraws


@pytest.mark.gpu
def test_device_differential_evolution_refuses_what_it_cannot_do():
    s = capi.HipSampler(64, 4, capi.CALC_ISO_GAUSSIAN, None, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    pos = po.init_positions(po.F64, 64, 4)
    s.set_state(pos, s.calc_logp(pos))
    with pytest.raises(capi.HipError):
        s.seek(10)
    with pytest.raises(capi.HipError):
        capi.HipSampler(64, 4, capi.CALC_ISO_GAUSSIAN, None, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION, shard_begin=0, shard_count=8)
    with pytest.raises(capi.HipError):
        capi.HipSampler(64, 4, capi.CALC_ISO_GAUSSIAN, None, mover=7)
