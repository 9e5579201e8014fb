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
