"""Several independent ensembles ("chains") stepped by the same launches (mcmcpp_hip_config.num_chains; BASELINE config 4
-- 8 independent 16 384-walker chains -- on one GPU).  Chain k must be, bit for bit, what a sampler of its own with seed
seed + k computes: the reference's EnsembleSampler objects one after the other (EnsembleSampler.h:199-218)."""
import numpy as np
import pytest

from mcmcpp_amd import capi, workloads
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _oracles(K, W, D, calc, params, seed):
    out = []
    for k in range(K):
        orc = po.Oracle(W, D, calc, params, seed=seed + k)
        pos = po.init_positions(po.F64, W, D, salt=10 + k)
        logp = orc.logp(pos)
        orc.set_state(pos, logp)
        out.append((orc, pos, logp))
    return out


@pytest.mark.parametrize("K,W,D,calc", [(3, 4096, 32, po.CALC_DENSE_GAUSSIAN), (2, 2048, 16, po.CALC_ISO_GAUSSIAN), (5, 1024, 7, po.CALC_ROSENBROCK),
                                        (16, 512, 32, po.CALC_DENSE_GAUSSIAN),
                                        # together more than 32 768 walkers: stepped by the half-step kernels (matrix-core and plain)
                                        (5, 8192, 32, po.CALC_DENSE_GAUSSIAN), (3, 16384, 8, po.CALC_ISO_GAUSSIAN), (2, 32768, 16, po.CALC_ROSENBROCK)])
@pytest.mark.parametrize("pinned", [False, True])
def test_chains_stepped_together_equal_chains_stepped_alone(K, W, D, calc, pinned):
    rng = np.random.default_rng(5)
    params = None
    if calc == po.CALC_DENSE_GAUSSIAN:
        params = workloads.ar1_precision(D, 0.5).ravel() if D == 32 else (lambda a: (a @ a.T / D + np.eye(D)).ravel())(rng.standard_normal((D, D)))
    if calc == po.CALC_ROSENBROCK:
        params = np.array([1.0, 100.0, 0.05])
    seed = 7
    orcs = _oracles(K, W, D, calc, params, seed)
    hip = capi.HipSampler(W, D, calc, params, seed=seed, num_chains=K)
    hip.set_state(np.stack([o[1] for o in orcs]), np.stack([o[2] for o in orcs]))
    for n_saved, interval in ((3, 2), (1, 1), (2, 5)):  # (odd step counts: the live ensembles end in the second buffer)
        out = None
        if pinned:
            out = capi.pinned_empty((K, n_saved, W, D))
            out[:] = np.nan
        chain, acc = hip.run(n_saved, interval=interval, out=out)
        assert chain.shape == (K, n_saved, W, D) and acc.shape == (K, n_saved * interval)
        for k, (orc, _, _) in enumerate(orcs):
            want_chain, want_acc = orc.run(n_saved, interval=interval, mode=po.MODE_COUNTER, threads=4)
            np.testing.assert_array_equal(acc[k], want_acc, err_msg="chain %d" % k)
            np.testing.assert_array_equal(chain[k], want_chain, err_msg="chain %d" % k)
    pos, logp, nacc = hip.get_state()
    total = 0
    for k, (orc, _, _) in enumerate(orcs):
        opos, ologp, onacc = orc.get_state()
        np.testing.assert_array_equal(pos[k], opos)
        np.testing.assert_array_equal(logp[k], ologp)
        np.testing.assert_array_equal(nacc[k], onacc)
        total += int(onacc.sum())
    c = hip.counters()
    assert c["accepted"] == total and c["near_ties"] == 0 and c["redraws"] == 0 and c["ensemble_steps"] == 17
    # nothing stored, counters only, then a reset
    _, acc = hip.run(1, interval=4, save_chain=False)
    for k, (orc, _, _) in enumerate(orcs):
        _, want_acc = orc.run(1, interval=4, save_chain=False, mode=po.MODE_COUNTER, threads=4)
        np.testing.assert_array_equal(acc[k], want_acc)
    hip.reset_counters()
    assert hip.counters()["accepted"] == 0


def test_config4_eight_chains_of_c2_in_its_one_gpu_form():
    """BASELINE config 4 exactly, on one GPU: 8 independent chains of 16 384 walkers x 32 dims, correlated Gaussian,
    seeds 0..7 (the reference: eight EnsembleSampler objects, EnsembleSampler.h:199-218), stepped by the same launches.
    Chain 0 is C2 itself and must reproduce the reference's own fixture `c2_16384x32`; chains 1..7 are checked against the
    oracle."""
    from tests.goldens import Golden
    g = Golden("c2_16384x32")
    K, W, D = 8, g.W, g.D
    assert (W, D, g.seed, g.slicing) == (16384, 32, 0, 1)
    orcs = []
    for k in range(1, K):
        orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, g.params, seed=k)
        pos = po.init_positions(po.F64, W, D, salt=k)
        logp = orc.logp(pos)
        orc.set_state(pos, logp)
        orcs.append((orc, pos, logp))
    hip = capi.HipSampler(W, D, po.CALC_DENSE_GAUSSIAN, g.params, seed=0, num_chains=K)
    hip.set_state(np.stack([g.init_pos] + [o[1] for o in orcs]), np.stack([g.init_logp] + [o[2] for o in orcs]))
    chain, acc = hip.run(g.steps)
    for k in g.checked_steps:
        if k >= 1:
            g.check_chain_step(k, chain[0][k - 1])
    np.testing.assert_array_equal(acc[0], g.accepted_per_call)
    assert W + int(acc[0].sum()) == g.accepted_total  # (the reference counts the initial placement)
    for k, (orc, _, _) in enumerate(orcs, start=1):
        want_chain, want_acc = orc.run(g.steps, mode=po.MODE_COUNTER, threads=8)
        np.testing.assert_array_equal(acc[k], want_acc, err_msg="chain %d" % k)
        np.testing.assert_array_equal(chain[k], want_chain, err_msg="chain %d" % k)
    c = hip.counters()
    assert c["near_ties"] == 0 and c["redraws"] == 0 and c["ensemble_steps"] == g.steps


def test_chains_config_is_checked():
    with pytest.raises(capi.HipError):  # too many
        capi.HipSampler(512, 8, po.CALC_ISO_GAUSSIAN, None, num_chains=17)
    with pytest.raises(capi.HipError):  # one mover only
        capi.HipSampler(512, 8, po.CALC_ISO_GAUSSIAN, None, num_chains=2, mover=capi.MOVER_DIFFERENTIAL_EVOLUTION)
    h = capi.HipSampler(512, 8, po.CALC_ISO_GAUSSIAN, None, num_chains=2)
    with pytest.raises(capi.HipError):
        h.half_step_async(0)
