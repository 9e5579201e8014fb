"""Multi-GPU drivers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Two ways the path shards:

* independent chains (BASELINE config 4): nothing on the data path is exchanged; `aggregate_throughput`
  is the only collective -- the benchmark's max-over-ranks time and sum of work.
* one ensemble split over the ranks (BASELINE config 5): `SplitEnsemble`.  Every rank keeps a full replica of
  the positions, updates its slice of the current colour, and the just-updated rows are all-gathered in place
  before the next half-step (the complementary half must be complete: any walker may pick any partner).
  Random draws are addressed by the global walker index, so the trajectory does not depend on the number of
  ranks and equals the single-GPU / reference one.

The stepping itself is delegated to a *shard backend*: `HipShardBackend` (the product: libmcmcpp_hip.so on
this rank's GPU, positions in a torch CUDA tensor shared with RCCL) or any object with the same five methods
(the CPU tests drive the same code with an oracle-based backend over gloo).
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


def aggregate_throughput(elapsed_s, work_units, extra_sums=(), group=None, device=None):
    """Whole-job throughput of a weak-scaling run: (sum of work over ranks) / (max time over ranks).

    Returns (throughput, max_elapsed, total_work, [summed extras]).  Works without a process group (N = 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return work_units / elapsed_s, elapsed_s, work_units, list(extra_sums)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    s = torch.tensor([work_units] + list(extra_sums), dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    vals = [float(x) for x in s.tolist()]
    return vals[0] / float(t.item()), float(t.item()), vals[0], vals[1:]


def shard_bounds(n_half, world, rank):
    """Walkers [begin, begin+count) of each colour owned by `rank`; equal contiguous slices (all-gather needs them equal)."""
    if n_half % world:
        raise ValueError("W/2 = %d is not divisible by the number of ranks %d" % (n_half, world))
    count = n_half // world
    return rank * count, count


class HipShardBackend:
    """This rank's slice on its GPU, for callers that bring their own exchange (the product path is a handle with an
    RCCL communicator: capi.HipSampler(comm_world=...), whose run() enqueues launches and exchanges itself).  Positions
    live in a torch CUDA tensor handed to libmcmcpp_hip.so as `device_positions`; launches and the collectives issued
    through torch.distributed go to ONE dedicated non-blocking stream (not torch's legacy default stream, which would
    serialise with every blocking stream of the process), so they are ordered with each other and with nothing else."""

    def __init__(self, W, D, calc_id, params, seed, stream, dtype, begin, count, device):
        from . import capi
        self.capi = capi
        t = torch.float64 if dtype == capi.F64 else torch.float32
        self.pos = torch.empty((W, D), dtype=t, device=device)
        self.stream = torch.cuda.Stream(device=self.pos.device)
        self.sampler = capi.HipSampler(W, D, calc_id, params, seed=seed, stream=stream, dtype=dtype,
                                       device=self.pos.device.index, shard_begin=begin, shard_count=count,
                                       device_positions=self.pos.data_ptr(), hip_stream=self.stream.cuda_stream)

    def set_state(self, pos, logp):
        self.sampler.set_state(pos, logp)

    def half_step(self, color):
        self.sampler.half_step_async(color)

    def exchange_context(self):
        """Collectives of the exchange are issued inside this context: on the stream the half-steps run on."""
        return torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    def positions(self):
        return self.pos

    def local_state(self):
        pos, logp, nacc = self.sampler.get_state()
        return logp, nacc

    def diagnostics(self):
        c = self.sampler.counters()
        return c["near_ties"], c["redraws"]


class SplitEnsemble:
    """One ensemble of W walkers split over the ranks of a process group."""

    def __init__(self, W, D, backend_factory, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.W, self.D, self.n = W, D, W // 2
        self.begin, self.count = shard_bounds(self.n, self.world, self.rank)
        self.backend = backend_factory(self.begin, self.count)
        self.half_steps = 0

    def set_state(self, pos, logp):
        """Every rank passes the same full arrays (as with the single-GPU sampler)."""
        self.backend.set_state(pos, logp)
        self.half_steps = 0

    def _exchange(self, color):
        pos = self.backend.positions()
        half = pos[color * self.n:(color + 1) * self.n]
        mine = half[self.begin:self.begin + self.count]
        # in place: the input is this rank's slice of the output (ncclAllGather's in-place form); on the backend's stream
        ctx = self.backend.exchange_context() if hasattr(self.backend, "exchange_context") else contextlib.nullcontext()
        with ctx:
            dist.all_gather_into_tensor(half.reshape(-1), mine.reshape(-1), group=self.group)

    def step(self):
        """One ensemble step: red half-step, exchange, black half-step, exchange (EnsembleSampler.h:341-354)."""
        for color in (0, 1):
            self.backend.half_step(color)
            self._exchange(color)
            self.half_steps += 1

    def run(self, n_saved, interval=1, save_chain=True):
        """n_saved*interval ensemble steps; the last of every `interval` is kept.  Returns the chain
        [n_saved, W, D] as a tensor on the positions' device (every rank holds the full replica)."""
        pos = self.backend.positions()
        chain = torch.empty((n_saved, self.W, self.D), dtype=pos.dtype, device=pos.device) if save_chain else None
        ctx = self.backend.exchange_context() if hasattr(self.backend, "exchange_context") else contextlib.nullcontext()
        for k in range(n_saved):
            for _ in range(interval):
                self.step()
            if chain is not None:
                with ctx:
                    chain[k].copy_(pos)
        if hasattr(self.backend, "synchronize"):
            self.backend.synchronize()
        return chain

    def gather_state(self):
        """Full (positions, logp, n_accept) on every rank."""
        if hasattr(self.backend, "synchronize"):
            self.backend.synchronize()
        logp, nacc = self.backend.local_state()
        pos = self.backend.positions()
        logp_t = torch.as_tensor(np.ascontiguousarray(logp)).to(pos.device)
        nacc_t = torch.as_tensor(np.ascontiguousarray(nacc).astype(np.int64)).to(pos.device)
        for color in (0, 1):
            for t in (logp_t, nacc_t):
                half = t[color * self.n:(color + 1) * self.n]
                dist.all_gather_into_tensor(half, half[self.begin:self.begin + self.count].clone(), group=self.group)
        return pos.clone(), logp_t, nacc_t

    def diagnostics(self):
        ties, redraws = self.backend.diagnostics()
        t = torch.tensor([ties, redraws], dtype=torch.int64, device=self.backend.positions().device)
        dist.all_reduce(t, group=self.group)
        return int(t[0]), int(t[1])
