#!/usr/bin/env python3
"""bench.py -- walker-steps/s of the stretch-move ensemble step on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 16 384 walkers x 32 dims, correlated Gaussian
log-posterior (Sigma_ij = 0.5^|i-j|, dense 32x32 precision matrix), fp64, StretchMove, seed = rank.
One bench "step" = one runMCMC batch of `--batch` ensemble steps (default 2000, C2's run length: 20 stored steps at
slicing interval 100, stored steps downloaded to the host inside the timed region; everything else stays
resident in HBM).  For N > 1 every rank runs its own independent chain of the same size on its own GPU
(BASELINE.json configs[3]: no data-path collective; weak scaling); the value is the sum over ranks
divided by the slowest rank's time.

The JSON line also carries
  roofline      achieved HBM bytes/s of the dominant kernel: algorithmic bytes per step-kernel launch
                ((2D+1)*8 read + (D+1)*8 written per walker update = 784 B at D = 32, x the
                walker updates one launch performs: all 16 384 with the full-step kernel this workload takes,
                8 192 with the half-step kernels) / the average launch duration measured with HIP events on
                the launch stream around the replayed launches (mcmcpp_hip_last_run_timing)
  cpu_baseline  the reference itself (oracle/_ref, kind "reference") or the oracle ("port") timed on this
                host's cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def ar1_precision(D, rho):
    from mcmcpp_amd import workloads
    return workloads.ar1_precision(D, rho)


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when its first communicator comes up; the driver reads ONE JSON line
    from stdout, so the process group is brought up with file descriptor 1 pointed at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_baseline(W, D, P, sample_steps):
    """Time the CPU path on a bounded sample of the workload.  oracle/ is imported here and only here: as the
    measured baseline, never by the GPU path."""
    from oracle import pyoracle as po
    # the GPU box gives one GPU job a share of 16 host cores (gpurun notes); more threads only thrash
    cores = min(len(os.sched_getaffinity(0)) or 1, 16)
    pos = po.init_positions(po.F64, W, D, salt=0)
    orc = po.Oracle(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), seed=0)
    logp = orc.logp(pos)
    out = {}
    if po.reference_available():
        # MCMC::EnsembleSampler of the reference, one thread, slicing so that only one step is stored
        r = po.reference_run(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), 0, pos, logp, 1, 1, slicing=sample_steps,
                             want_chain=False)
        out = dict(value=W * sample_steps / r["seconds"], unit="walker-steps/s", cores=1, kind="reference",
                   sample="MCMC::EnsembleSampler<double, StretchMove> (reference, 1 thread): %d ensemble steps of "
                          "the %dx%d workload, %.1f s" % (sample_steps, W, D, r["seconds"]))
    # the oracle on all cores (static walker partition): the fair analogue of ParallelEnsembleSampler
    orc.set_state(pos, logp)
    orc.run(1, interval=2, save_chain=False, mode=po.MODE_COUNTER, threads=cores)  # start the thread pool
    orc.set_state(pos, logp)
    t0 = time.perf_counter()
    orc.run(1, interval=sample_steps, save_chain=False, mode=po.MODE_COUNTER, threads=cores)
    dt = time.perf_counter() - t0
    port = dict(value=W * sample_steps / dt, unit="walker-steps/s", cores=cores, kind="port",
                sample="oracle, %d threads: %d ensemble steps of the %dx%d workload, %.1f s"
                       % (cores, sample_steps, W, D, dt))
    if not out:
        out = port
    else:
        out["all_cores_port"] = port
    return out


def bench_split(args, rank, local_rank, world, dist, torch, capi):
    """BASELINE config 5: one 131 072-walker x 64-dim isotropic-Gaussian ensemble over all ranks; strong scaling."""
    from mcmcpp_amd import distributed as md
    from mcmcpp_amd import workloads
    if dist is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            dist.barrier()
    W, D = 131072, 64
    steps_per = 50
    dev = "cuda:%d" % local_rank
    ens = md.SplitEnsemble(W, D, lambda b, c: md.HipShardBackend(W, D, capi.CALC_ISO_GAUSSIAN, None, 0, 0, capi.F64, b, c, dev))
    pos = workloads.init_positions(W, D, salt=0)
    logp = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, device=local_rank).calc_logp(pos)
    ens.set_state(pos, logp)
    for _ in range(args.warmup):
        ens.run(1, interval=steps_per, save_chain=False)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ens.run(1, interval=steps_per, save_chain=False)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    _, _, nacc = ens.gather_state()
    accepted = int(nacc.sum().item())
    total = float(W) * steps_per * (args.steps + args.warmup)
    if rank == 0:
        exchanged = (W // 2) * D * 8 * (world - 1) / world  # bytes each rank receives per half-step
        print(json.dumps({
            "metric": "walker-steps/sec + acceptance rate, 131072 walkers x 64 dims split over the GPUs",
            "value": float(W) * steps_per * args.steps / elapsed, "unit": "walker-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "acceptance_rate": accepted / total,
            "config": {"workload": "C5: one 131072x64 isotropic-Gaussian ensemble split over %d GPU(s), RCCL all-gather "
                                   "of the updated half-ensemble slice after every half-step; one step = %d ensemble steps"
                                   % (world, steps_per), "walkers": W, "dims": D},
            "allgather": {"bytes_received_per_rank_per_half_step": exchanged,
                          "half_steps": 2 * steps_per * args.steps}}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2000, help="ensemble steps per bench step (SURVEY.md 8d: C2 runs 2 000 steps)")
    ap.add_argument("--interval", type=int, default=100, help="slicing interval (one stored step per interval)")
    ap.add_argument("--walkers", type=int, default=16384)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--cpu-sample-steps", type=int, default=1000,
                    help="ensemble steps of the workload timed on the CPU (about 20 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="experiments only: store nothing")
    ap.add_argument("--no-accepted", action="store_true", help="experiments only: skip the per-step accepted counters")
    ap.add_argument("--mode", default="chains", choices=["chains", "split"],
                    help="chains (default, the headline): one independent C2 chain per GPU; split: BASELINE config 5, "
                         "one 131072x64 ensemble split over the GPUs with an RCCL all-gather per half-step")
    ap.add_argument("--calc", default="dense", choices=["dense", "iso", "rosenbrock"],
                    help="experiments only: the headline workload is the dense (correlated) Gaussian")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal only (MCMCPP_BENCH_BACKEND=gloo): several ranks may share the GPUs of a smaller box; the driver's
    # multi-GPU runs use the default, RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("MCMCPP_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("%d ranks but %d GPUs visible" % (world, torch.cuda.device_count()))
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            dist.barrier()  # brings the communicator up now
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    from mcmcpp_amd import capi, workloads

    if args.mode == "split":
        return bench_split(args, rank, local_rank, world, dist, torch, capi)

    W, D = args.walkers, args.dims
    assert args.batch % args.interval == 0
    n_saved = args.batch // args.interval
    P = ar1_precision(D, 0.5)
    calc_id, calc_params = {"dense": (capi.CALC_DENSE_GAUSSIAN, P.ravel()), "iso": (capi.CALC_ISO_GAUSSIAN, None),
                            "rosenbrock": (capi.CALC_ROSENBROCK, [1.0, 100.0, 0.05])}[args.calc]
    sampler = capi.HipSampler(W, D, calc_id, calc_params, seed=rank, device=local_rank)
    pos = workloads.init_positions(W, D, salt=rank)
    sampler.set_state(pos, sampler.calc_logp(pos))

    # the stored steps land in host memory that already exists, like a block of the facade's Chain (allocated
    # once, filled as the run proceeds); the download itself is inside the timed region
    chain_block = None if args.no_chain else np.zeros((n_saved, W, D))

    def bench_step():
        return sampler.run(n_saved, interval=args.interval, save_chain=not args.no_chain,
                           want_accepted=not args.no_accepted, out=chain_block)

    for _ in range(args.warmup):
        bench_step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    accepted = 0
    gpu_ms = 0.0
    launches = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, acc = bench_step()
        accepted += int(acc.sum()) if acc is not None else 0
        ms, nl = sampler.last_run_timing()
        gpu_ms += ms
        launches += nl
    barrier()
    elapsed = time.perf_counter() - t0

    from mcmcpp_amd import distributed as md
    walker_steps = float(W) * args.batch * args.steps
    # whole-job aggregate: sum of the ranks' work / slowest rank's time (tests/test_split_gloo.py covers this helper)
    _, elapsed, walker_steps, (accepted, gpu_ms, launches) = md.aggregate_throughput(
        elapsed, walker_steps, [float(accepted), gpu_ms, float(launches)], device=reduce_device)

    if rank == 0:
        bytes_per_update = (2 * D + 1) * 8 + (D + 1) * 8          # 784 B at D = 32, fp64
        saved_fraction = 1.0 / args.interval
        # walker updates one launch performs: W/2 for the half-step kernels, W when the library steps with one
        # launch per ensemble step (full_step_kernel.hpp)
        updates_per_launch = walker_steps / launches
        full_step = updates_per_launch > 0.75 * W
        kernel = ("stretch_full_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16>" if full_step else
                  "stretch_half_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16, P=2>")
        bytes_per_launch = updates_per_launch * (bytes_per_update + saved_fraction * D * 8)
        us_per_launch = gpu_ms * 1e3 / launches
        achieved = bytes_per_launch / (us_per_launch * 1e-6) / 1e9
        # HBM bytes per launch from the PMC counters: collected by a separate rocprofv3 --pmc run of this very
        # command (counters cannot be read from inside the process) and committed with its provenance
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if args.calc == "dense" and (W, D) == (16384, 32) and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("kernel", "").split("<")[0] == kernel.split("<")[0]:
                traffic = pmc["hbm_bytes_per_launch"]
        line = {
            "metric": "walker-steps/sec + acceptance rate, 16384 walkers x 32 dims, 1/2/4/8 GPU",
            "value": walker_steps / elapsed,
            "unit": "walker-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "acceptance_rate": accepted / walker_steps,
            "config": {"workload": "C2: %d walkers x %d dims, correlated Gaussian (rho=0.5, dense precision), "
                                   "StretchMove, fp64; one step = %d ensemble steps, 1 stored step per %d "
                                   "downloaded; %s" % (W, D, args.batch, args.interval,
                                                       "one independent chain per GPU (seed = rank)" if world > 1
                                                       else "one chain"),
                       "walkers": W, "dims": D, "ensemble_steps_per_step": args.batch,
                       "slicing_interval": args.interval, "chains": world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel,
                         "walker_updates_per_launch": updates_per_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": us_per_launch,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                           "passes; FETCH_SIZE doubled per the gfx950 correction)" if traffic else None,
                         "note": "avg launch duration = HIP-event time on the launch stream over the graph-replayed "
                                 "step launches / launches; it includes the ~1.5 us dependent-launch boundary"},
        }
        if args.calc != "dense":
            line["config"]["workload"] += " [EXPERIMENT: calculator = %s, not the headline workload]" % args.calc
        if world == 1 and not args.no_cpu_baseline and args.calc == "dense":
            line["cpu_baseline"] = cpu_baseline(W, D, P, args.cpu_sample_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
