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
