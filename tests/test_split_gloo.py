"""The N > 1 paths on the CPU (gloo, world_size 2): the split-ensemble driver (mcmcpp_amd/distributed.py)
exchanging half-ensembles between ranks, and the benchmark's whole-job aggregation.  The stepping backend
here is built on the oracle (tests may use it); on GPUs the same driver code runs HipShardBackend over RCCL."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcmcpp_amd import distributed as md
from oracle import pyoracle as po


class OracleShardBackend:
    def __init__(self, W, D, calc, params, seed, begin, count):
        self.orc = po.Oracle(W, D, calc, params, seed=seed)
        self.begin, self.count = begin, count
        self.pos = torch.from_numpy(self.orc.positions_view())  # zero-copy: the exchange writes the oracle's array

    def set_state(self, pos, logp):
        self.orc.set_state(pos, logp)

    def half_step(self, color):
        self.orc.half_step_shard(color, self.begin, self.count)
        self.orc.half_step_commit()

    def positions(self):
        return self.pos

    def local_state(self):
        _, logp, nacc = self.orc.get_state()
        return logp, nacc

    def diagnostics(self):
        return self.orc.near_ties, self.orc.redraws


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, D, steps = 256, 8, 12
        params = np.array([1.0, 100.0, 0.05])
        pos = po.init_positions(po.F64, W, D, salt=4)
        logp = po.Oracle(W, D, po.CALC_ROSENBROCK, params).logp(pos)
        ens = md.SplitEnsemble(W, D, lambda b, c: OracleShardBackend(W, D, po.CALC_ROSENBROCK, params, 7, b, c))
        assert (ens.begin, ens.count) == (rank * 64, 64)
        ens.set_state(pos, logp)
        chain = ens.run(steps // 2, interval=2)
        fpos, flogp, fnacc = ens.gather_state()
        ties, redraws = ens.diagnostics()
        thr, tmax, total, extras = md.aggregate_throughput(1.0 + rank, 100.0 * (rank + 1), [float(rank + 1)])
        if rank == 0:
            np.savez(out, chain=chain.numpy(), pos=fpos.numpy(), logp=flogp.numpy(), nacc=fnacc.numpy(),
                     ties=ties, redraws=redraws, thr=thr, tmax=tmax, total=total, extra=extras[0])
    finally:
        dist.destroy_process_group()


def test_split_ensemble_two_ranks_equals_single_chain(tmp_path):
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, 29000 + os.getpid() % 2000, out), nprocs=2, join=True)
    z = np.load(out)
    W, D, steps = 256, 8, 12
    params = np.array([1.0, 100.0, 0.05])
    ref = po.Oracle(W, D, po.CALC_ROSENBROCK, params, seed=7)
    pos = po.init_positions(po.F64, W, D, salt=4)
    ref.set_state(pos, ref.logp(pos))
    chain, acc = ref.run(steps // 2, interval=2)
    rpos, rlogp, rnacc = ref.get_state()
    np.testing.assert_array_equal(z["chain"], chain)      # the trajectory does not depend on the rank count
    np.testing.assert_array_equal(z["pos"], rpos)
    np.testing.assert_array_equal(z["logp"], rlogp)
    np.testing.assert_array_equal(z["nacc"], rnacc)
    assert int(z["ties"]) == 0 and int(z["redraws"]) == 0
    # whole-job aggregation: sum of work / slowest rank
    assert float(z["tmax"]) == 2.0 and float(z["total"]) == 300.0 and float(z["thr"]) == 150.0 and float(z["extra"]) == 3.0


def test_shard_bounds():
    assert md.shard_bounds(65536, 8, 3) == (3 * 8192, 8192)
    with pytest.raises(ValueError):
        md.shard_bounds(50, 4, 0)


def test_aggregate_without_process_group():
    thr, tmax, total, extras = md.aggregate_throughput(2.0, 10.0, [3.0])
    assert (thr, tmax, total, extras) == (5.0, 2.0, 10.0, [3.0])
