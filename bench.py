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
  cpu_baseline  the reference itself (oracle/_ref, kind "reference": built at the reference's own optimisation level,
                oracle/Makefile) or the oracle ("port") timed on this host's cores on a bounded sample of the same
                workload (rank 0, N = 1 only)
  secondary     (N = 1) the other single-GPU configurations of BASELINE.json, each with its own roofline object:
                C3 (65 536 x 32 Rosenbrock), config 5's ensemble (131 072 x 64) and config 4's eight chains on one GPU,
                and C2's target under Mover::DifferentialEvolution
`--mode split` is BASELINE config 5 proper: one ensemble split over the ranks, launches and RCCL exchanges enqueued by
libmcmcpp_hip.so itself (mcmcpp_hip_config.comm_*).
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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa_node(torch, device, sysfs="/sys"):
    """Several ranks on one node: keep this rank's threads (and so the pages of its chain block and the pinned ring its
    GPU writes stored steps into) on the NUMA node its GPU hangs off.  Returns a note for the JSON line; any doubt (no
    PCI address, no node, fewer than 8 permitted cores there) leaves the affinity alone."""
    try:
        p = torch.cuda.get_device_properties(device)
        addr = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")).read())
        if node < 0:
            return "unchanged (GPU %s reports no NUMA node)" % addr
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
        allowed = cpus & os.sched_getaffinity(0)
        if len(allowed) < 8:
            return "unchanged (%d permitted cores on NUMA node %d of GPU %s)" % (len(allowed), node, addr)
        os.sched_setaffinity(0, allowed)
        return "NUMA node %d of GPU %s (%d cores)" % (node, addr, len(allowed))
    except Exception as e:  # noqa: BLE001 -- placement is an optimisation, never a reason to fail the bench
        return "unchanged (%s: %s)" % (type(e).__name__, e)


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
        # MCMC::EnsembleSampler of the reference, one thread, slicing so that only one step is stored; the build at the
        # reference's own optimisation level when it is there (-O3, contraction on: timing only, never parity)
        timed = po.timed_reference_available()
        r = po.reference_run(W, D, po.CALC_DENSE_GAUSSIAN, P.ravel(), 0, pos, logp, 1, 1, slicing=sample_steps,
                             want_chain=False, timed_build=timed)
        out = dict(value=W * sample_steps / r["seconds"], unit="walker-steps/s", cores=1, kind="reference",
                   sample="MCMC::EnsembleSampler<double, StretchMove> (reference, 1 thread, %s): %d ensemble steps of "
                          "the %dx%d workload, %.1f s" % ("-O3 -march=x86-64-v3 build" if timed else "-O2 contraction-off parity build",
                                                          sample_steps, W, D, r["seconds"]))
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


def source_digest():
    """SHA-256 over the kernel and host sources of the library: what a counter measurement is valid for."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mcmcpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def counter_traffic(kernel, secondary_key=None):
    """HBM bytes per launch of `kernel` from the newest committed counter file (profiles/r*_pmc_traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc passes of this very command) -- only if it was taken from the
    sources this library was built from; otherwise null: a stale number is worse than none.  secondary_key: one of the
    secondary configurations' step kernels, recorded in the same file under "secondary"."""
    import glob
    digest = source_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except ValueError:
            continue
        if pmc.get("source_sha256") != digest:
            continue
        if secondary_key is not None:
            rec = pmc.get("secondary", {}).get(secondary_key)
            if rec is not None and rec.get("kernel", "").split("<")[0] == kernel.split("<")[0]:
                return rec["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        elif pmc.get("kernel", "").split("<")[0] == kernel.split("<")[0]:
            return pmc["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def residency_note(W, D, elem, chains=1, buffers=1):
    """Where the walkers live between two launches: ensembles below the 256 MiB of Infinity Cache never reach HBM, so their
    roofline fraction (algorithmic bytes over the HBM peak) is a NOMINAL figure -- it can exceed what HBM itself delivers
    (about 6.3 of the 8 TB/s, MI355X_MICROARCH.md) and says how far the launch is from the bandwidth bound, not from HBM."""
    mib = W * D * elem * chains * buffers / 2.0**20
    if mib < 256:
        return "positions %.0f MiB: resident in Infinity Cache (256 MiB); the fraction of the 8 TB/s HBM peak is nominal" % mib
    return "positions %.0f MiB: beyond Infinity Cache, served by HBM" % mib


def live_counter_traffic(kernel_sub, seconds_per_pass=60.0):
    """HBM bytes per launch of the headline step kernel measured IN THIS RUN: two child processes, each this very script
    (2 bench steps, nothing else) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- separate passes, the
    program itself behind `--`, never combined with another trace domain, as MI355X_MICROARCH.md prescribes; FETCH_SIZE
    doubled (gfx950 counts 128-byte read requests at 64 bytes), both counters are KiB per dispatch.  Runs behind the timed
    region.  Returns (bytes per launch, note) or (None, why): a missing profiler, a pass that fails or overruns its time
    limit (the child is killed by its own process id) simply leaves the committed figure in place."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself being profiled"
    means = {}
    tmp = tempfile.mkdtemp(prefix="mcmcpp_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-live-counters"]
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                rc = child.wait(timeout=seconds_per_pass)
            except subprocess.TimeoutExpired:
                child.kill()
                child.wait()
                return None, "the %s pass overran %.0f s" % (counter, seconds_per_pass)
            if rc != 0:
                return None, "the %s pass ended with code %d" % (counter, rc)
            vals = []
            for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(path)):
                    if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, "no %s rows for %s" % (counter, kernel_sub)
            means[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0]) * 1024.0
    return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over "
                   "`bench.py --steps 2` (%d / %d dispatches of the step kernel); 2 x FETCH_SIZE + WRITE_SIZE, KiB per dispatch, "
                   "FETCH_SIZE doubled per the gfx950 correction" % (means["FETCH_SIZE"][1], means["WRITE_SIZE"][1]))


def roofline_of(W, D, walker_steps, launches, gpu_ms, kernel, saved_fraction=0.0, rows_read=2, elem=8, chains=1):
    """(2D+1)*s read + (D+1)*s written per walker update (+ D*s when the step is stored), s = element size
    (8: fp64, 4: fp32), times the walker updates one launch performs, over the average launch duration (HIP events on the
    launch stream).  rows_read = 3: Mover::DifferentialEvolution, whose update reads two partner rows."""
    bytes_per_update = (rows_read * D + 1) * elem + (D + 1) * elem
    updates_per_launch = walker_steps / launches
    bytes_per_launch = updates_per_launch * (bytes_per_update + saved_fraction * D * elem)
    us_per_launch = gpu_ms * 1e3 / launches
    achieved = bytes_per_launch / (us_per_launch * 1e-6) / 1e9
    full_step = updates_per_launch > 0.75 * W * chains
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "kernel": kernel, "walker_updates_per_launch": updates_per_launch,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": us_per_launch,
            "residency": residency_note(W, D, elem, chains, 2 if full_step else 1)}


def secondary_config(capi, workloads, device, name, W, D, calc, params, kernel, batch, seconds=1.0, chains=1, mover=None, traffic_key=None, dtype=0):
    """One of the other single-GPU configurations: runs of `batch` ensemble steps (nothing stored) for about `seconds`,
    with its roofline.  chains > 1: that many independent ensembles (seeds 0, 1, ...) stepped by the same launches.
    dtype: capi.F64 (0) / capi.F32 (1), the ParamType of the reference's templates."""
    if mover is None:
        s = capi.HipSampler(W, D, calc, params, seed=0, device=device, num_chains=chains, dtype=dtype)
    else:
        s = capi.HipSampler(W, D, calc, params, seed=0, device=device, mover=mover, dtype=dtype)
    pos = np.stack([workloads.init_positions(W, D, salt=k) for k in range(chains)]) if chains > 1 else workloads.init_positions(W, D, salt=0)
    s.set_state(pos, s.calc_logp(pos))
    s.run(1, interval=batch, save_chain=False)
    accepted = 0
    gpu_ms, launches, reps = 0.0, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, acc = s.run(1, interval=batch, save_chain=False)
        accepted += int(acc.sum())
        ms, nl = s.last_run_timing()
        gpu_ms += ms
        launches += nl
        reps += 1
    dt = time.perf_counter() - t0
    ws = float(W) * chains * batch * reps
    s.close()
    roof = roofline_of(W, D, ws, launches, gpu_ms, kernel, rows_read=3 if mover is not None else 2, elem=4 if dtype == 1 else 8, chains=chains)
    if traffic_key is not None:
        roof["traffic"], roof["traffic_source"] = counter_traffic(kernel, traffic_key)
    return {"workload": name, "walkers": W, "dims": D, "chains": chains, "dtype": "f32" if dtype == 1 else "f64", "value": ws / dt, "unit": "walker-steps/s",
            "ensemble_steps": batch * reps, "seconds": dt, "acceptance_rate": accepted / ws, "roofline": roof}


LOOPBACK_LIB = os.path.join(ROOT, "tests", "cpp", "_build", "libloopback_ccl.so")


SPLIT_STEPS_PER = 50  # ensemble steps per bench step of the split measurement


def split_rank_numbers(args, capi, torch, local_rank, G, ranks_here, cid, steps, warmup, sync):
    """Steps C5's ensemble as rank(s) `ranks_here` of G (one rank: this process is that rank; several: loop-back ranks as threads
    of this process) and returns {rank: raw numbers}.  sync(): what brackets the timed loop besides the library's own
    collectives (which keep the ranks in lockstep anyway)."""
    import threading
    from mcmcpp_amd import workloads
    W, D = args.split_walkers, 64
    steps_per = SPLIT_STEPS_PER
    pos = workloads.init_positions(W, D, salt=0)
    probe = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, device=local_rank)
    logp = probe.calc_logp(pos)
    probe.close()
    threaded = len(ranks_here) > 1
    thread_barrier = threading.Barrier(len(ranks_here)) if threaded else None
    results = {}

    def one_rank(r):
        with (_Nothing() if threaded else _StdoutToStderr()):  # (RCCL's banner)
            ens = capi.HipSampler(W, D, capi.CALC_ISO_GAUSSIAN, None, seed=0, device=local_rank, comm_world=G, comm_rank=r, comm_id=cid)
        ens.set_state(pos, logp)
        for _ in range(warmup):
            ens.run(1, interval=steps_per, save_chain=False)
        accepted, gpu_ms, launches, enq_ms, xchg_us, xbytes, repeats, slots = 0, 0.0, 0, 0.0, 0.0, 0.0, 0, 0
        if thread_barrier is not None:
            thread_barrier.wait()
        else:
            sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, acc = ens.run(1, interval=steps_per, save_chain=False)
            accepted += int(acc.sum())  # (ensemble-wide counts: all-reduced by the library)
            ms, nl = ens.last_run_timing()
            gpu_ms += ms
            launches += nl
            e, _, x = ens.last_run_host_timing()
            enq_ms += e
            xchg_us += x
            b, rp, sl = ens.last_run_exchange()
            xbytes += b
            repeats += rp
            slots = sl
        if thread_barrier is not None:
            thread_barrier.wait()
        else:
            sync()
        results[r] = dict(elapsed=time.perf_counter() - t0, accepted=accepted, gpu_ms=gpu_ms, launches=launches, enq_ms=enq_ms,
                          xchg_us=xchg_us, xbytes=xbytes / steps, repeats=repeats, slots=slots)
        ens.close()

    if threaded:
        errs = []

        def guarded(r):
            try:
                one_rank(r)
            except Exception as e:  # noqa: BLE001
                errs.append("rank %d: %s" % (r, e))
                thread_barrier.abort()
        threads = [threading.Thread(target=guarded, args=(r,)) for r in ranks_here]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise RuntimeError("; ".join(errs))
    else:
        one_rank(ranks_here[0])
    return results


def split_measurement(args, rank, local_rank, world, dist, torch, capi, steps, warmup, loopback_ranks=0):
    """BASELINE config 5: one 131 072-walker x 64-dim isotropic-Gaussian ensemble over all ranks; strong scaling.
    Every rank owns one handle with an RCCL communicator: mcmcpp_hip_run enqueues the step launches and the exchanges.
    Returns the measurement (rank 0; None elsewhere).

    loopback_ranks = G > 0 (rehearsals on a one-GPU box; MCMCPP_HIP_RCCL_LIB must name tests/cpp/loopback_ccl's library before
    the library binds its collectives): G ranks as G threads of THIS process, all on this process's GPU -- the product's
    world > 1 code with a loop-back exchange; throughput figures then describe G ranks sharing one GPU, not a G-GPU job."""
    G = loopback_ranks if loopback_ranks > 0 else world
    with _StdoutToStderr():  # (RCCL's banner)
        if world > 1 and loopback_ranks == 0:
            box = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            cid = box[0]
        else:
            cid = capi.comm_unique_id()

    def outer_barrier():
        if dist is not None and loopback_ranks == 0:
            dist.barrier()
        torch.cuda.synchronize()

    ranks_here = list(range(G)) if loopback_ranks > 0 else [rank]
    results = split_rank_numbers(args, capi, torch, local_rank, G, ranks_here, cid, steps, warmup, outer_barrier)
    mine = results[ranks_here[0]]
    elapsed = mine["elapsed"]
    if dist is not None and loopback_ranks == 0:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    return split_line(args, world, G, mine, elapsed, steps, warmup, loopback_ranks)


def split_line(args, world, G, mine, elapsed, steps, warmup, loopback_ranks):
    """The bench line of a split measurement from one rank's raw numbers and the slowest rank's time."""
    W, D = args.split_walkers, 64
    steps_per = SPLIT_STEPS_PER
    n_steps = steps_per * steps
    total = float(W) * n_steps
    launches, gpu_ms = mine["launches"], mine["gpu_ms"]
    per_step_launches = launches / n_steps  # 1: one exchange per ensemble step; 2: one per half-step
    whole = W * D * 8 * (G - 1) / G  # position bytes a rank receives per ensemble step when whole slices are all-gathered
    recv = mine["xbytes"]
    xchg = mine["xchg_us"] / steps
    # what one rank updates per launch: its slice of one colour (half-step kernels) or of both (full-step kernels)
    updates = float(W) / G / (2 if per_step_launches > 1.5 else 1)
    kernel = ("stretch_full_step_kernel<double, IsoGaussianFn, EPL=2, LPW=32>" if per_step_launches < 1.5 else
              "stretch_half_step_kernel<double, IsoGaussianFn, EPL=2, LPW=32>")
    step_us = gpu_ms * 1e3 / n_steps  # GPU time of one ensemble step on the launch stream: kernels + exchange
    # per launch, between several ranks without the sampled exchange time; a single rank's "exchange" moves nothing and
    # its events only bracket launch gaps, so there the whole stream time is charged to the launches
    kern_us = (max(step_us - xchg, 1e-9) if G > 1 else step_us) / per_step_launches
    bytes_per_update = (2 * D + 1) * 8 + (D + 1) * 8
    achieved = updates * bytes_per_update / (kern_us * 1e-6) / 1e9
    how = ("%d GPU(s)" % world) if loopback_ranks == 0 else ("%d ranks as threads of one process on ONE GPU, exchanging through the test-only "
                                                              "loop-back collective library (a rehearsal of the code path, not a %d-GPU measurement)" % (G, G))
    return {
        "metric": "walker-steps/sec + acceptance rate, 131072 walkers x 64 dims split over the GPUs",
        "value": total / elapsed, "unit": "walker-steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "acceptance_rate": mine["accepted"] / total,
        "config": {"workload": "C5: one %dx%d isotropic-Gaussian ensemble split over %s; launches and exchanges "
                               "enqueued by libmcmcpp_hip.so (%s); one step = %d ensemble steps"
                               % (W, D, how, "one exchange per ensemble step, full-step kernels on the rank's slice"
                                  if per_step_launches < 1.5 else "one exchange per half-step", steps_per),
                   "walkers": W, "dims": D, "ranks": G, "cpu_affinity_rank0": args.cpu_affinity},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": kernel, "walker_updates_per_launch": updates,
                     "algorithmic_bytes_per_launch": updates * bytes_per_update, "avg_launch_us": kern_us,
                     "note": "per rank; launch time = (stream time of a step - sampled exchange time) / launches per step; with one "
                             "rank the exchange calls are charged to the launches"},
        "allgather": {"bytes_received_per_rank_per_step": recv, "bytes_if_whole_slices_were_gathered": whole,
                      "block_slots": mine["slots"], "repeated_chunks": mine["repeats"],
                      "exchange_us_per_step": xchg,
                      "gb_per_s_per_rank": (recv / (xchg * 1e-6) / 1e9) if xchg > 0 and G > 1 else None,
                      "gb_per_s_per_link": (recv / (G - 1) / (xchg * 1e-6) / 1e9) if xchg > 0 and G > 1 else None,
                      "link_peak_gb_per_s": 153.0,
                      "note": "ranks of more than one exchange only the rows that moved (mcmcpp_hip_last_run_exchange: packed blocks, one "
                              "all-gather per exchange, scatter; block_slots = walkers per block learned from the run, 0 = whole slices); "
                              "exchange time = HIP events around a sample of exchanges (pack + all-gather + scatter) on the launch stream; "
                              "a single rank moves nothing (rates null)"},
        "host": {"enqueue_us_per_step": mine["enq_ms"] * 1e3 / n_steps, "gpu_us_per_step": step_us,
                 "note": "time the host thread spends enqueueing one ensemble step (launches + exchange calls) against the "
                         "GPU time of that step: the host must stay below it"},
    }


def split_leg_in_children(args, rank, local_rank, world, dist, torch, capi, steps, warmup):
    """The C5 leg of a multi-rank run, every rank of it in a CHILD process of the bench rank that owns the GPU: a crash or a hang
    of code that has never met RCCL between GPUs before then costs this secondary entry, not the headline
    the parent has already measured.  The parents hand the communicator id down, wait (bounded), collect the children's raw
    numbers, agree on success and on the slowest rank's time; rank 0 assembles the line."""
    import subprocess
    with _StdoutToStderr():
        box = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
    cid = box[0]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "split", "--child-rank", str(rank), "--child-world", str(world),
           "--comm-id-hex", bytes(cid).hex(), "--child-device", str(local_rank), "--steps", str(steps), "--warmup", str(warmup),
           "--split-walkers", str(args.split_walkers), "--no-live-counters"]
    mine, why = None, ""
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=max(5.0, args.split_leg_seconds - 5.0))
        if child.returncode == 0:
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            mine = json.loads(lines[-1]) if lines else None
            why = "" if mine else "the child printed no numbers"
        else:
            why = "the child of rank %d ended with code %d: %s" % (rank, child.returncode, err.strip()[-300:])
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        why = "the child of rank %d did not finish in time" % rank
    ok = torch.tensor([1.0 if mine else 0.0, mine["elapsed"] if mine else 0.0], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    oks = [torch.zeros_like(ok) for _ in range(world)]
    dist.all_gather(oks, ok)
    if not all(float(o[0]) > 0 for o in oks):
        return {"workload": "C5 split over the ranks", "error": why or "another rank's child failed"} if rank == 0 else None
    if rank != 0:
        return None
    elapsed = max(float(o[1]) for o in oks)
    return split_line(args, world, world, mine, elapsed, steps, warmup, 0)


class _Nothing:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def bench_split(args, rank, local_rank, world, dist, torch, capi):
    line = split_measurement(args, rank, local_rank, world, dist, torch, capi, args.steps, args.warmup, loopback_ranks=args.loopback_ranks)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed bench steps (default 500: about 6 s of stepping, so that the timed region dominates the run "
                         "and a GPU-busy sampler sees it; --mode split: 20)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2000, help="ensemble steps per bench step (SURVEY.md 8d: C2 runs 2 000 steps)")
    ap.add_argument("--interval", type=int, default=100, help="slicing interval (one stored step per interval)")
    ap.add_argument("--walkers", type=int, default=16384)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--cpu-sample-steps", type=int, default=2000,
                    help="ensemble steps of the workload timed on the CPU (about 20 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C5-ensemble secondary measurements")
    ap.add_argument("--secondary-seconds", type=float, default=1.0, help="stepping time of each secondary configuration (counter passes use less)")
    ap.add_argument("--no-chain", action="store_true", help="experiments only: store nothing")
    ap.add_argument("--pinned-chain", action="store_true",
                    help="stored steps into a pinned block from the library (MCMCPP_CHAIN_MEMORY=pinned in the facade): the launches "
                         "write them in place, no staging ring, no host copy")
    ap.add_argument("--no-accepted", action="store_true", help="experiments only: skip the per-step accepted counters")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc child passes behind the timed region); "
                         "the committed counter file is reported instead while it matches the sources")
    ap.add_argument("--mode", default="chains", choices=["chains", "split"],
                    help="chains (default, the headline): one independent C2 chain per GPU; split: BASELINE config 5, "
                         "one 131072x64 ensemble split over the GPUs, exchanged over RCCL by the library")
    ap.add_argument("--split-walkers", type=int, default=131072,
                    help="--mode split: walkers of the ensemble (16384 on one GPU = what one rank of the 8-GPU job updates)")
    ap.add_argument("--loopback-ranks", type=int, default=0,
                    help="--mode split on ONE GPU: that many ranks as threads of this process, exchanging through the test-only loop-back "
                         "collective library (tests/cpp/loopback_ccl.hip) -- a rehearsal of the world > 1 code path")
    ap.add_argument("--child-rank", type=int, default=-1,
                    help="internal (--mode split): this process is one rank of --child-world ranks of the split measurement, started by "
                         "a rank of a multi-rank run; it prints its raw numbers as one JSON object")
    ap.add_argument("--child-world", type=int, default=0)
    ap.add_argument("--comm-id-hex", default="")
    ap.add_argument("--child-device", type=int, default=0)
    ap.add_argument("--no-split-leg", action="store_true", help="several ranks: skip the bounded C5 (split ensemble) measurement behind the chains")
    ap.add_argument("--split-leg-seconds", type=float, default=45.0, help="several ranks: time limit of that measurement")
    ap.add_argument("--calc", default="dense", choices=["dense", "iso", "rosenbrock"],
                    help="experiments only: the headline workload is the dense (correlated) Gaussian")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.mode == "split" else 500

    if args.loopback_ranks > 0 or os.environ.get("MCMCPP_BENCH_SPLIT_LEG") == "loopback":
        # (before the library binds its collectives, i.e. before the first handle with a communicator)
        if not os.path.exists(LOOPBACK_LIB):
            raise SystemExit("%s is missing: python -m pytest tests/test_split_loopback.py -k exports builds it" % LOOPBACK_LIB)
        os.environ["MCMCPP_HIP_RCCL_LIB"] = LOOPBACK_LIB

    if args.child_rank >= 0:
        # One rank of a split measurement in a process of its own (see the several-ranks branch below): no process group --
        # the communicator id comes from the parent, the library's collectives keep the ranks in lockstep.
        import torch
        torch.cuda.set_device(args.child_device)
        from mcmcpp_amd import capi
        args.cpu_affinity = "inherited from the parent rank"
        res = split_rank_numbers(args, capi, torch, args.child_device, args.child_world, [args.child_rank], bytes.fromhex(args.comm_id_hex),
                                 args.steps, args.warmup, torch.cuda.synchronize)
        print(json.dumps(res[args.child_rank]), flush=True)
        return

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
    # (before any library thread exists: the handle's worker threads inherit the affinity)
    want_bind = os.environ.get("MCMCPP_BENCH_NUMA_BIND", "1" if world > 1 else "0") == "1"
    args.cpu_affinity = bind_to_gpu_numa_node(torch, local_rank) if want_bind else "unchanged (one rank, or MCMCPP_BENCH_NUMA_BIND=0)"
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

    # The stored steps land in host memory that already exists, like a block of the facade's Chain
    # (include/MCMCpp/Device/SamplerCore.h): by default heap memory, as the facade's default -- the launches forward stored
    # steps to a pinned staging ring and the host copies them out while the launches continue; with --pinned-chain a pinned
    # block from the library, which the launches write in place.  The transfer is inside the timed region either way.
    if args.no_chain:
        chain_block = None
    elif args.pinned_chain:
        chain_block = capi.pinned_empty((n_saved, W, D))
        chain_block[:] = 0.0
    else:
        chain_block = np.zeros((n_saved, W, D))

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
    per_rank_values = None
    if dist is not None:
        # every rank's own throughput beside the aggregate (the aggregate divides by the slowest rank's time)
        mine = torch.tensor([walker_steps / elapsed], dtype=torch.float64, device=reduce_device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank_values = [float(v.item()) for v in everyone]
    # whole-job aggregate: sum of the ranks' work / slowest rank's time (tests/test_split_gloo.py covers this helper)
    _, elapsed, walker_steps, (accepted, gpu_ms, launches) = md.aggregate_throughput(
        elapsed, walker_steps, [float(accepted), gpu_ms, float(launches)], device=reduce_device)

    if rank == 0:
        saved_fraction = 0.0 if args.no_chain else 1.0 / args.interval
        # walker updates one launch performs: W/2 for the half-step kernels, W when the library steps with one
        # launch per ensemble step (full_step_kernel.hpp)
        full_step = walker_steps / launches > 0.75 * W
        kernel = ("stretch_full_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16>" if full_step else
                  "stretch_half_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16, P=2>")
        roof = roofline_of(W, D, walker_steps, launches, gpu_ms, kernel, saved_fraction)
        # HBM bytes per launch from the PMC counters: collected by separate rocprofv3 --pmc passes of this very command
        # (counters cannot be read from inside the process), committed with the digest of the sources they were taken
        # from, and reported only while that digest is the one this library was built from
        traffic, traffic_file = counter_traffic(kernel) if args.calc == "dense" and (W, D) == (16384, 32) else (None, None)
        roof["traffic"] = traffic
        roof["traffic_source"] = (traffic_file + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                                  "the gfx950 correction; taken from the sources this library was built from)") if traffic else None
        roof["note"] = ("avg launch duration = HIP-event time on the launch stream over the graph-replayed step launches / launches; "
                        "it includes the ~1.5 us dependent-launch boundary; the working set lives in Infinity Cache (see residency), so "
                        "frac is a nominal fraction of the HBM peak: the launch is latency-bound, not bandwidth-bound")
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
                       "slicing_interval": args.interval, "chains": world, "cpu_affinity_rank0": args.cpu_affinity,
                       "chain_memory": "none" if args.no_chain else ("pinned block from the library" if args.pinned_chain else "heap block (pageable)")},
            "roofline": roof,
        }
        if per_rank_values is not None:
            line["per_rank_values"] = per_rank_values
        if args.calc != "dense":
            line["config"]["workload"] += " [EXPERIMENT: calculator = %s, not the headline workload]" % args.calc
        if world == 1 and not args.no_secondary and args.calc == "dense":
            sampler.close()
            line["secondary"] = [
                secondary_config(capi, workloads, local_rank, "C3: 65536 walkers x 32 dims, Rosenbrock log-posterior, StretchMove, fp64; "
                                 "runs of 1000 ensemble steps, nothing stored", 65536, 32, capi.CALC_ROSENBROCK, [1.0, 100.0, 0.05],
                                 "stretch_half_step_kernel<double, RosenbrockFn, EPL=2, LPW=16>", 1000, traffic_key="C3", seconds=args.secondary_seconds),
                secondary_config(capi, workloads, local_rank, "C5's ensemble on ONE GPU: 131072 walkers x 64 dims, isotropic Gaussian, "
                                 "StretchMove, fp64; runs of 500 ensemble steps, nothing stored", 131072, 64, capi.CALC_ISO_GAUSSIAN, None,
                                 "stretch_half_step_kernel<double, IsoGaussianFn, EPL=2, LPW=32>", 500, traffic_key="C5_one_gpu", seconds=args.secondary_seconds),
                secondary_config(capi, workloads, local_rank, "C4 on ONE GPU: 8 independent chains (seeds 0..7) of 16384 walkers x 32 dims, correlated "
                                 "Gaussian, StretchMove, fp64, stepped by the same launches (mcmcpp_hip_config.num_chains = 8); runs of 2000 "
                                 "ensemble steps, nothing stored", 16384, 32, capi.CALC_DENSE_GAUSSIAN, P.ravel(),
                                 "stretch_half_step_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16, P=4, several chains, late draws>", 2000, chains=8, traffic_key="C4_one_gpu", seconds=args.secondary_seconds),
                secondary_config(capi, workloads, local_rank, "C2's target under the other ensemble mover (SURVEY 8f row f3): 16384 walkers x 32 dims, "
                                 "correlated Gaussian, Mover::DifferentialEvolution, fp64; runs of 2000 ensemble steps, nothing stored; the launch "
                                 "time includes the stream-planning launches", 16384, 32, capi.CALC_DENSE_GAUSSIAN, P.ravel(),
                                 "de_update_mfma_kernel<double, DenseGaussianFn, EPL=2, LPW=16> (+ de_boundary_kernel, one per 64 half-steps)", 2000,
                                 mover=capi.MOVER_DIFFERENTIAL_EVOLUTION, traffic_key="DE_C2", seconds=args.secondary_seconds),
                # ParamType = float: the same two shapes at half the bytes per element
                secondary_config(capi, workloads, local_rank, "C2's target in fp32 (SURVEY 8f row f4): 16384 walkers x 32 dims, correlated Gaussian, StretchMove, "
                                 "ParamType = float; runs of 2000 ensemble steps, nothing stored", 16384, 32, capi.CALC_DENSE_GAUSSIAN, P.ravel(),
                                 "stretch_full_step_mfma_kernel<float, DenseGaussianFn, EPL=2, LPW=16>", 2000, traffic_key="C2_f32", seconds=args.secondary_seconds, dtype=capi.F32),
                secondary_config(capi, workloads, local_rank, "C5's ensemble in fp32 on ONE GPU: 131072 walkers x 64 dims, isotropic Gaussian, StretchMove, "
                                 "ParamType = float; runs of 500 ensemble steps, nothing stored", 131072, 64, capi.CALC_ISO_GAUSSIAN, None,
                                 "stretch_half_step_kernel<float, IsoGaussianFn, EPL=4, LPW=16>", 500, traffic_key="C5_one_gpu_f32", seconds=args.secondary_seconds, dtype=capi.F32),
            ]
        if world == 1 and not args.no_cpu_baseline and args.calc == "dense":
            line["cpu_baseline"] = cpu_baseline(W, D, P, args.cpu_sample_steps)
        if world == 1 and not args.no_live_counters and args.calc == "dense" and (W, D) == (16384, 32) and not args.no_chain:
            # the counters of THIS run (the committed file stays as the fallback and as the secondary lines' source)
            sampler.close()
            torch.cuda.synchronize()
            live, note = live_counter_traffic(kernel.split("<")[0])
            if live is not None:
                roof["traffic_committed"] = roof["traffic"]
                roof["traffic"] = live
                roof["traffic_source"] = note
            else:
                roof["traffic_live"] = "not taken (%s); the committed counter file is reported" % note
    else:
        line = None

    if world > 1 and not args.no_split_leg and args.calc == "dense":
        # Several ranks: the same ranks also step BASELINE config 5 -- ONE ensemble split between them, exchanged by the
        # library -- for a bounded time, and the result rides along as a secondary measurement.  The headline above is
        # complete at this point: a watchdog prints it without the split object and ends the process (exit, never
        # re-exec) if the leg has not come back in time, e.g. with a rank stuck in a collective.
        import threading
        sampler.close()
        done = threading.Event()

        def watchdog():
            if done.wait(args.split_leg_seconds):
                return
            if rank == 0:
                line.setdefault("secondary", []).append({"workload": "C5 split over the ranks", "error": "did not finish within %.0f s; "
                                                         "the headline above was complete before it started" % args.split_leg_seconds})
                print(json.dumps(line), flush=True)
            sys.stdout.flush()
            os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            loop = world if os.environ.get("MCMCPP_BENCH_SPLIT_LEG") == "loopback" else 0
            if loop:
                # rehearsal: rank 0 alone steps all ranks as threads through the loop-back library
                split = split_measurement(args, rank, local_rank, world, dist, torch, capi, steps=10, warmup=2, loopback_ranks=loop) if rank == 0 else None
                if dist is not None:
                    dist.barrier()
            else:
                split = split_leg_in_children(args, rank, local_rank, world, dist, torch, capi, steps=10, warmup=2)
            broke_here = False
        except Exception as e:  # noqa: BLE001 -- the headline must not be lost to the secondary measurement
            split = {"workload": "C5 split over the ranks", "error": "%s: %s" % (type(e).__name__, e)}
            broke_here = True
        done.set()
        if rank == 0 and split is not None:
            split["workload"] = split.get("config", {}).get("workload", split.get("workload"))
            line.setdefault("secondary", []).append(split)
        if broke_here:
            # (an exception in THIS rank's part of the leg: the other ranks may be waiting in a collective this rank never
            #  joined -- their watchdogs end them; a tidy shutdown of the process group would wait for them)
            if rank == 0:
                print(json.dumps(line), flush=True)
            sys.stdout.flush()
            os._exit(0)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
