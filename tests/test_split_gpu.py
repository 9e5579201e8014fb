"""GPU tests of the split-ensemble path on one MI355X: sharded handles (global random-stream addressing,
caller-owned position replica, caller's stream) and the torch.distributed driver with HipShardBackend.
The 8-GPU RCCL exchange itself cannot run on the one-GPU box; the driver code around it is the same that
tests/test_split_gloo.py runs with two ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mcmcpp_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("W,D,calc,shards", [(4096, 32, po.CALC_ISO_GAUSSIAN, 2), (2048, 64, po.CALC_DENSE_GAUSSIAN, 4),
                                             (1024, 7, po.CALC_ROSENBROCK, 8)])
def test_sharded_handles_sharing_one_replica_follow_the_single_chain(W, D, calc, shards):
    import torch
    rng = np.random.default_rng(3)
    params = None
    if calc == po.CALC_DENSE_GAUSSIAN:
        a = rng.standard_normal((D, D))
        params = (a @ a.T / D + np.eye(D)).ravel()
    if calc == po.CALC_ROSENBROCK:
        params = np.array([1.0, 100.0, 0.05])
    n = W // 2
    orc = po.Oracle(W, D, calc, params, seed=5)
    pos = po.init_positions(po.F64, W, D, salt=2)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    replica = torch.empty((W, D), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    handles = [capi.HipSampler(W, D, calc, params, seed=5, shard_begin=k * (n // shards), shard_count=n // shards,
                               device_positions=replica.data_ptr(), hip_stream=stream) for k in range(shards)]
    for h in handles:
        assert h.device_positions() == replica.data_ptr()
        assert h.shard_span(1) == ((n + h.cfg.shard_begin) * D, (n // shards) * D)
        h.set_state(pos, logp)
    steps = 9
    want_chain, want_acc = orc.run(steps)
    for s in range(steps):
        for color in (0, 1):
            for h in handles:  # every shard of the colour, then (no-op here) the exchange
                h.half_step_async(color)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(replica.cpu().numpy(), want_chain[s])
    opos, ologp, onacc = orc.get_state()
    for k, h in enumerate(handles):
        _, hl, hn = h.get_state()
        for color in (0, 1):
            sl = slice(color * n + k * (n // shards), color * n + (k + 1) * (n // shards))
            np.testing.assert_array_equal(hl[sl], ologp[sl])
            np.testing.assert_array_equal(hn[sl], onacc[sl])
        c = h.counters()
        assert c["near_ties"] == 0 and c["redraws"] == 0 and c["ensemble_steps"] == steps
    assert sum(h.counters()["accepted"] for h in handles) == int(want_acc.sum())
    with pytest.raises(capi.HipError):
        handles[0].half_step_async(1)  # colours must alternate red, black, ...
    with pytest.raises(capi.HipError):
        handles[0].run(1)              # a sharded handle is not driven by run()


def _params(calc, D, rng):
    if calc == po.CALC_DENSE_GAUSSIAN:
        a = rng.standard_normal((D, D))
        return (a @ a.T / D + np.eye(D)).ravel()
    if calc == po.CALC_ROSENBROCK:
        return np.array([1.0, 100.0, 0.05])
    return None


@pytest.mark.parametrize("W,D,calc", [(4096, 32, po.CALC_ISO_GAUSSIAN), (4096, 32, po.CALC_DENSE_GAUSSIAN), (2048, 64, po.CALC_DENSE_GAUSSIAN),
                                      (1024, 7, po.CALC_ROSENBROCK)])
@pytest.mark.parametrize("scheme", ["exchange_per_step", "exchange_per_half_step"])
def test_split_ensemble_run_through_the_c_abi(monkeypatch, W, D, calc, scheme):
    """mcmcpp_hip_run on a handle with a REAL RCCL communicator (one rank: all this box has): launches and ncclAllGather /
    ncclAllReduce calls enqueued by the library itself, stored steps, ensemble-wide accepted counts, get_state -- bit for
    bit the oracle's chain.  More than one rank (sliced kernels, exchange offsets): tests/test_split_loopback.py."""
    monkeypatch.setenv("MCMCPP_HIP_COMM_FULL_STEP", "1" if scheme == "exchange_per_step" else "0")
    rng = np.random.default_rng(11)
    params = _params(calc, D, rng)
    orc = po.Oracle(W, D, calc, params, seed=9)
    pos = po.init_positions(po.F64, W, D, salt=1)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    hip = capi.HipSampler(W, D, calc, params, seed=9, comm_world=1, comm_rank=0, comm_id=capi.comm_unique_id())
    hip.set_state(pos, logp)
    for n_saved, interval in ((3, 2), (1, 1), (2, 3)):  # (odd step counts: the live ensemble ends in the second buffer)
        want_chain, want_acc = orc.run(n_saved, interval=interval)
        chain, acc = hip.run(n_saved, interval=interval)
        np.testing.assert_array_equal(acc, want_acc)
        np.testing.assert_array_equal(chain, want_chain)
        for got, want in zip(hip.get_state(), orc.get_state()):
            np.testing.assert_array_equal(got, want)
    c = hip.counters()
    assert c["near_ties"] == 0 and c["redraws"] == 0 and c["ensemble_steps"] == 13
    enq_ms, wall_ms, xchg_us = hip.last_run_host_timing()
    assert 0 < enq_ms <= wall_ms and xchg_us >= 0


def test_split_ensemble_config_is_checked():
    cid = capi.comm_unique_id()
    with pytest.raises(capi.HipError):  # W/2 must divide by the number of ranks
        capi.HipSampler(4098, 8, po.CALC_ISO_GAUSSIAN, None, comm_world=2, comm_rank=0, comm_id=cid)
    with pytest.raises(capi.HipError):  # rank outside the communicator
        capi.HipSampler(4096, 8, po.CALC_ISO_GAUSSIAN, None, comm_world=2, comm_rank=2, comm_id=cid)
    with pytest.raises(capi.HipError):  # a shard that is not the rank's slice
        capi.HipSampler(4096, 8, po.CALC_ISO_GAUSSIAN, None, comm_world=1, comm_rank=0, comm_id=cid, shard_begin=0, shard_count=512)
    with pytest.raises(capi.HipError):  # neither an id nor a communicator
        capi.HipSampler(4096, 8, po.CALC_ISO_GAUSSIAN, None, comm_world=1, comm_rank=0)


def test_config5_full_size_split_run_single_rank():
    """BASELINE config 5's ensemble (131 072 x 64) through the split path of the C ABI with one rank of a real RCCL
    communicator: 3 steps against the multi-threaded oracle (half-step scheme: the slice is the whole half).  Eight
    ranks of 8 192 walkers per colour: tests/test_split_loopback.py."""
    W, D = 131072, 64
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=0)
    pos = po.init_positions(po.F64, W, D, salt=0)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    want_chain, want_acc = orc.run(3, mode=po.MODE_COUNTER, threads=8)
    hip = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=0, comm_world=1, comm_rank=0, comm_id=capi.comm_unique_id())
    hip.set_state(pos, logp)
    chain, acc = hip.run(3)
    np.testing.assert_array_equal(acc, want_acc)
    np.testing.assert_array_equal(chain, want_chain)
    for got, want in zip(hip.get_state(), orc.get_state()):
        np.testing.assert_array_equal(got, want)
    hip.close()


def test_whole_ensemble_handle_on_the_legacy_default_stream():
    """hip_stream = NULL with MCMCPP_HIP_FLAG_CALLER_STREAM is the legacy default stream, on which HIP cannot capture
    graphs: run() must fall back to plain launches instead of failing."""
    W, D = 2048, 8
    orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=2)
    pos = po.init_positions(po.F64, W, D, salt=4)
    logp = orc.logp(pos)
    orc.set_state(pos, logp)
    hip = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=2, hip_stream=0)
    hip.set_state(pos, logp)
    want_chain, want_acc = orc.run(5, interval=2)
    chain, acc = hip.run(5, interval=2)
    np.testing.assert_array_equal(acc, want_acc)
    np.testing.assert_array_equal(chain, want_chain)


_TWO_RANKS = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from mcmcpp_amd import capi
from oracle import pyoracle as po
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
if rank == 0:
    open(idfile + ".tmp", "wb").write(capi.comm_unique_id()); os.rename(idfile + ".tmp", idfile)
else:
    while not os.path.exists(idfile): time.sleep(0.05)
cid = open(idfile, "rb").read()
W, D = 8192, 64
orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=3)
pos = po.init_positions(po.F64, W, D, salt=1); logp = orc.logp(pos); orc.set_state(pos, logp)
for scheme in ("1", "0"):
    os.environ["MCMCPP_HIP_COMM_FULL_STEP"] = scheme
    orc.set_state(pos, logp)
    want, want_acc = orc.run(5, interval=3, mode=po.MODE_COUNTER, threads=4)
    hip = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=3, device=rank, comm_world=world, comm_rank=rank, comm_id=cid)
    hip.set_state(pos, logp)
    chain, acc = hip.run(5, interval=3)
    assert np.array_equal(acc, want_acc), "accepted counts differ"
    assert np.array_equal(chain, want), "chains differ"
    for got, wanted in zip(hip.get_state(), orc.get_state()):
        assert np.array_equal(got, wanted)
    c = hip.counters()
    assert c["near_ties"] == 0 and c["redraws"] == 0
    hip.close()
print("SPLIT_RANK_%%d_OK" %% rank)
'''


def test_split_ensemble_over_two_gpus_through_the_c_abi(tmp_path):
    """One ensemble split over TWO GPUs, exchanged over RCCL by the library (both schemes), against the oracle.  Needs a box
    with two GPUs: the driver's one-GPU test box skips it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    idfile = str(tmp_path / "rccl_id")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANKS % {"root": ROOT}, str(r), "2", idfile], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (out, err) in enumerate(outs):
        assert "SPLIT_RANK_%d_OK" % r in out, out[-2000:] + err[-4000:]


_SINGLE_RANK = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mcmcpp_amd import capi, distributed as md
from oracle import pyoracle as po
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, D = 8192, 64
orc = po.Oracle(W, D, po.CALC_ISO_GAUSSIAN, None, seed=0)
pos = po.init_positions(po.F64, W, D); logp = orc.logp(pos); orc.set_state(pos, logp)
want, _ = orc.run(4, interval=2, mode=po.MODE_COUNTER, threads=4)
ens = md.SplitEnsemble(W, D, lambda b, c: md.HipShardBackend(W, D, capi.CALC_ISO_GAUSSIAN, None, 0, 0, capi.F64, b, c, "cuda:0"))
ens.set_state(pos, logp)
chain = ens.run(4, interval=2)
fpos, flogp, fnacc = ens.gather_state()
opos, ologp, onacc = orc.get_state()
assert np.array_equal(chain.cpu().numpy(), want)
assert np.array_equal(fpos.cpu().numpy(), opos) and np.array_equal(flogp.cpu().numpy(), ologp)
assert np.array_equal(fnacc.cpu().numpy(), onacc.astype(np.int64))
assert ens.diagnostics() == (0, 0)
dist.destroy_process_group()
print("SPLIT_SINGLE_RANK_OK")
'''


def test_split_driver_with_hip_backend_over_rccl_single_rank():
    out = subprocess.run([sys.executable, "-c", _SINGLE_RANK % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert "SPLIT_SINGLE_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
