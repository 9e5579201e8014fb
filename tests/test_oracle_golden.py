"""CPU tests: the oracle (oracle/stretch_oracle.c) against the reference's golden vectors, the pcg64
restatement against numpy's independent PCG64, and -- where oracle/_ref was prebuilt -- against the
compiled reference itself."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.goldens import DIGEST, GOLDEN_DIR, SMALL, Golden


def test_pcg32_known_answer_published_demo():
    # pcg32 (64-bit LCG, XSH-RR) seeded (42, 54): first outputs printed by the PCG paper's demo program.
    # Restated here in pure Python to pin the family's seeding rule (state = (seed+inc)*M+inc).
    M, mask = 6364136223846793005, 2**64 - 1
    inc = ((54 << 1) | 1) & mask
    st = ((42 + inc) * M + inc) & mask
    out = []
    for _ in range(6):
        old = st
        st = (old * M + inc) & mask
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        r = old >> 59
        out.append(((x >> r) | (x << ((-r) & 31))) & 0xFFFFFFFF)
    assert out == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


@pytest.mark.parametrize("seed,stream", [(0, 0), (42, 54), (2**63 + 5, 2**64 - 1), (-1 & (2**64 - 1), 7)])
def test_pcg64_matches_numpy(seed, stream):
    g = po.Pcg64(seed, stream)
    # seeding rule of pcg-cpp's engine(state, stream) (restated in Python)
    M = (2549297995355413924 << 64) | 4865540595714422341
    inc = ((stream << 1) | 1) & (2**128 - 1)
    assert g.inc == inc and g.state == ((seed + inc) * M + inc) & (2**128 - 1)
    bg = np.random.PCG64()
    st = bg.state
    st["state"] = {"state": g.state, "inc": g.inc}
    bg.state = st
    ours = [g.next() for _ in range(64)]
    theirs = [int(x) for x in bg.random_raw(64)]
    assert ours == theirs


@pytest.mark.parametrize("delta", [0, 1, 3, 24576, 3 * 8192 * 12345 + 17, 2**64 + 12345, 2**127 + 1])
def test_pcg64_advance_equals_stepping_and_numpy(delta):
    g = po.Pcg64(7, 3)
    bg = np.random.PCG64()
    st = bg.state
    st["state"] = {"state": g.state, "inc": g.inc}
    bg.state = st
    g.advance(delta)
    bg.advance(delta)
    assert g.state == bg.state["state"]["state"]
    if delta <= 24576:
        h = po.Pcg64(7, 3)
        for _ in range(delta):
            h.next()
        assert h.state == g.state
    m, p = po.jump_coeffs(g.inc, delta)
    h = po.Pcg64(7, 3)
    assert (m * h.state + p) & (2**128 - 1) == g.state


def test_canonical_conversions():
    L = po.lib()
    assert L.so_canonical_f64(0) == 0.0
    assert L.so_canonical_f64(2**63) == 0.5
    assert L.so_canonical_f64(2**64 - 1) == np.nextafter(1.0, 0.0)  # rounds to 2^64 -> clamped
    assert L.so_canonical_f64((2**53 - 1) << 11) == (2**53 - 1) / 2**53
    assert L.so_canonical_f32(2**64 - 1) == np.nextafter(np.float32(1), np.float32(0))
    assert L.so_canonical_f32(2**63) == 0.5
    r = 0x123456789ABCDEF1
    assert L.so_canonical_f64(r) == float(np.float64(r) / np.float64(2**64))


def _run_oracle(g, mode, threads=1):
    orc = po.Oracle(g.W, g.D, g.calc, g.params, seed=g.seed, dtype=g.dtype, alpha=g.alpha)
    np.testing.assert_array_equal(orc.logp(g.init_pos), g.init_logp)
    orc.set_state(g.init_pos, g.init_logp)
    done = 0
    acc_calls = []
    for k in sorted(set(g.checked_steps + [g.steps])):
        chain, acc = orc.run(k - done, interval=g.slicing, save_chain=True, mode=mode, threads=threads)
        acc_calls.append(acc.reshape(k - done, g.slicing).sum(axis=1))
        done = k
        if k in g.checked_steps:
            g.check_chain_step(k, chain[-1])
    acc_calls = np.concatenate(acc_calls)
    np.testing.assert_array_equal(acc_calls, g.accepted_per_call)
    pos, logp, nacc = orc.get_state()
    assert int(nacc.sum()) + g.W == g.accepted_total
    assert g.total_steps == g.W * (1 + g.steps * g.slicing)
    assert orc.near_ties == 0, "a decision was within a few ulp of flipping: pick another fixture seed"
    assert orc.redraws == 0
    return pos, logp, nacc


@pytest.mark.parametrize("name", SMALL + DIGEST)
def test_oracle_sequential_matches_reference_golden(name):
    _run_oracle(Golden(name), po.MODE_SEQUENTIAL)


@pytest.mark.parametrize("name", SMALL + DIGEST)
def test_oracle_counter_addressed_matches_reference_golden(name):
    g = Golden(name)
    a = _run_oracle(g, po.MODE_COUNTER, threads=1)
    b = _run_oracle(g, po.MODE_COUNTER, threads=3)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_oracle_resume_is_seamless():
    g = Golden("rosen80x8")
    one = po.Oracle(g.W, g.D, g.calc, g.params, seed=g.seed)
    one.set_state(g.init_pos, g.init_logp)
    chain_a, acc_a = one.run(60)
    two = po.Oracle(g.W, g.D, g.calc, g.params, seed=g.seed)
    two.set_state(g.init_pos, g.init_logp)
    c1, a1 = two.run(25, mode=po.MODE_COUNTER)
    c2, a2 = two.run(35, mode=po.MODE_SEQUENTIAL)
    np.testing.assert_array_equal(chain_a, np.concatenate([c1, c2]))
    np.testing.assert_array_equal(acc_a, np.concatenate([a1, a2]))


def test_oracle_reproduces_reference_own_test_run():
    """The reference's SkewedGaussian/StretchMove test (320 x 2, slicing 30, 40 019 stored steps) end to
    end: accepted/total as the reference prints them."""
    want = json.load(open(os.path.join(GOLDEN_DIR, "reference_skewed_test.json")))
    g = Golden("skewed320x2")  # same initial placement
    orc = po.Oracle(g.W, g.D, g.calc, g.params, seed=0)
    orc.set_state(g.init_pos, g.init_logp)
    _, acc = orc.run(want["stored_steps"], interval=want["slicing"], save_chain=False, mode=po.MODE_COUNTER,
                     threads=4)
    assert int(acc.sum()) + g.W == want["accepted_total"]
    assert g.W * (1 + want["stored_steps"] * want["slicing"]) == want["total_steps"]


def test_skewed_calculator_restatement_equals_reference_class():
    if not po.reference_available():
        pytest.skip("oracle/_ref not prebuilt here")
    g = Golden("skewed320x2")
    a = po.reference_run(g.W, g.D, po.CALC_SKEWED_GAUSSIAN_2D, g.params, 0, g.init_pos, g.init_logp, 50, 1)
    b = po.reference_run(g.W, g.D, 103, g.params, 0, g.init_pos, g.init_logp, 50, 1)  # this repo's class
    np.testing.assert_array_equal(a["chain"], b["chain"])


@pytest.mark.parametrize("calc,W,D,dtype", [(po.CALC_ISO_GAUSSIAN, 34, 3, po.F64), (po.CALC_ROSENBROCK, 66, 9, po.F32),
                                            (po.CALC_DENSE_GAUSSIAN, 130, 12, po.F64)])
def test_oracle_against_live_reference(calc, W, D, dtype):
    """Fresh (non-fixture) cases straight against the compiled reference, where it is available."""
    if not po.reference_available():
        pytest.skip("oracle/_ref not prebuilt here")
    t = po.np_dtype(dtype)
    rng = np.random.default_rng(5)
    params = None
    if calc == po.CALC_ROSENBROCK:
        params = np.array([1.0, 100.0, 0.05], dtype=t)
    if calc == po.CALC_DENSE_GAUSSIAN:
        a = rng.standard_normal((D, D))
        params = (a @ a.T / D + np.eye(D)).astype(t).ravel()
    orc = po.Oracle(W, D, calc, params, seed=11, dtype=dtype)
    pos = po.init_positions(dtype, W, D, salt=9)
    logp = orc.logp(pos)
    ref = po.reference_run(W, D, calc, params, 11, pos, logp, 1, 120, dtype=dtype)
    orc.set_state(pos, logp)
    chain, acc = orc.run(120)
    np.testing.assert_array_equal(ref["chain"][1:], chain)
    assert int(acc.sum()) + W == int(ref["accepted"][-1])
