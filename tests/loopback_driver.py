"""Runs the split ensemble with MORE THAN ONE RANK on one GPU: G handles on G host threads, each with its own replica,
exchanging through tests/cpp/loopback_ccl.hip (a loop-back stand-in for RCCL, selected with MCMCPP_HIP_RCCL_LIB).

The collective library is bound once per process, so tests/test_split_loopback.py starts this file as ONE subprocess with
the environment variable set; it runs every case and prints one JSON line per case.  What is compared, bit for bit,
against the oracle: the stored chain on every rank that asked for it, the ensemble-wide accepted counts per step (an
all-reduce) on every rank, get_state (positions, log-posteriors, per-walker counters) on every rank, near ties = redraws = 0.
"""
import json
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mcmcpp_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def _params(calc, D, rng):
    if calc == po.CALC_DENSE_GAUSSIAN:
        a = rng.standard_normal((D, D))
        return (a @ a.T / D + np.eye(D)).ravel()
    if calc == po.CALC_ROSENBROCK:
        return np.array([1.0, 100.0, 0.05])
    return None


def run_ranks(G, W, D, calc, dtype=capi.F64, scheme="step", runs=((3, 2), (1, 1), (2, 3)), chain_on="all", seed=9, env=None,
              oracle_threads=4, bad_rank=None, expect=None):
    """One ensemble over G ranks; returns a list of problems (empty: everything equal to the oracle).
    expect: optional check of the exchange statistics, called with [(bytes per step, repeated chunks, block slots)] of
    rank 0's runs; returns a list of problems."""
    env = dict(env or {})
    env["MCMCPP_HIP_COMM_FULL_STEP"] = "1" if scheme == "step" else "0"
    saved_env = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    problems = []
    try:
        rng = np.random.default_rng(11)
        params = _params(calc, D, rng)
        po_t = po.F64 if dtype == capi.F64 else po.F32
        orc = po.Oracle(W, D, calc, params, seed=seed, dtype=po_t)
        pos = po.init_positions(po_t, W, D, salt=1)
        logp = orc.logp(pos)
        orc.set_state(pos, logp)
        want = []
        for n_saved, interval in runs:
            chain, acc = orc.run(n_saved, interval=interval, mode=po.MODE_COUNTER, threads=oracle_threads)
            want.append((chain, acc, orc.get_state()))
        cid = capi.comm_unique_id()
        got = [None] * G
        errors = [None] * G

        def rank_main(r):
            try:
                hip = capi.HipSampler(W, D, calc, params, seed=seed, dtype=dtype, device=0, comm_world=G, comm_rank=r, comm_id=cid)
                hip.set_state(pos, logp)
                out = []
                if bad_rank is not None:
                    # one rank asks for something impossible: every rank must come back with an error, none may hang,
                    # and the ensemble must be untouched afterwards
                    try:
                        hip.run(2, interval=0 if r == bad_rank else 1)
                        out.append("no error")
                    except capi.HipError as e:
                        out.append("error %d" % e.code)
                for n_saved, interval in runs:
                    keep = chain_on == "all" or r == 0
                    # (only rank 0 and the last rank ask for the per-step accepted counts: the all-reduce behind them must not
                    #  depend on who asks)
                    wants = r == 0 or r == G - 1
                    chain, acc = hip.run(n_saved, interval=interval, save_chain=keep, want_accepted=wants)
                    out.append((chain, acc, hip.get_state(), hip.last_run_exchange()))
                out.append(hip.counters())
                out.append(hip.last_run_host_timing())
                got[r] = out
                hip.close()
            except Exception:
                errors[r] = traceback.format_exc()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for r in range(G):
            if errors[r]:
                problems.append("rank %d raised: %s" % (r, errors[r][-600:]))
                continue
            out = got[r]
            if bad_rank is not None:
                verdict = out.pop(0)
                if not verdict.startswith("error"):
                    problems.append("rank %d: the bad request of rank %d went unnoticed (%s)" % (r, bad_rank, verdict))
            for k, (n_saved, interval) in enumerate(runs):
                chain, acc, state, xstats = out[k]
                wchain, wacc, wstate = want[k]
                if acc is not None and not np.array_equal(acc, wacc):
                    problems.append("rank %d run %d: accepted counts differ" % (r, k))
                if chain is not None and not np.array_equal(chain, wchain):
                    problems.append("rank %d run %d: chains differ" % (r, k))
                for name, a, b in zip(("positions", "logp", "n_accept"), state, wstate):
                    if not np.array_equal(a, b):
                        problems.append("rank %d run %d: %s differ (%d cells)" % (r, k, name, int(np.sum(np.asarray(a) != np.asarray(b)))))
            c = out[len(runs)]
            if c["near_ties"] != 0 or c["redraws"] != 0:
                problems.append("rank %d: near ties %d, redraws %d" % (r, c["near_ties"], c["redraws"]))
            if c["ensemble_steps"] != sum(a * b for a, b in runs):
                problems.append("rank %d: %d ensemble steps counted" % (r, c["ensemble_steps"]))
        if not problems:
            # every rank took the same decisions about its exchange blocks
            for k in range(len(runs)):
                if len({got[r][k][3][1:] for r in range(G)}) != 1:
                    problems.append("run %d: the ranks disagree about repeated chunks / block slots: %r" % (k, [got[r][k][3] for r in range(G)]))
            if expect is not None:
                problems += expect([got[0][k][3] for k in range(len(runs))])
        # the ranks' own accepted totals add up to the ensemble's
        if not problems:
            total = sum(got[r][len(runs)]["accepted"] for r in range(G))
            if total != int(sum(int(w[1].sum()) for w in want)):
                problems.append("accepted totals of the ranks add up to %d" % total)
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return problems


CASES = {}


def case(name):
    def reg(fn):
        CASES[name] = fn
        return fn
    return reg


for _G in (2, 4, 8):
    for _scheme in ("step", "half"):
        case("iso8192x64_G%d_%s" % (_G, _scheme))(lambda G=_G, scheme=_scheme: run_ranks(G, 8192, 64, po.CALC_ISO_GAUSSIAN, scheme=scheme))
for _scheme in ("step", "half"):
    case("dense4096x32_G4_%s" % _scheme)(lambda scheme=_scheme: run_ranks(4, 4096, 32, po.CALC_DENSE_GAUSSIAN, scheme=scheme, chain_on="rank0"))
    case("rosen1024x7_G8_%s" % _scheme)(lambda scheme=_scheme: run_ranks(8, 1024, 7, po.CALC_ROSENBROCK, scheme=scheme))
    case("iso2048x16_f32_G2_%s" % _scheme)(lambda scheme=_scheme: run_ranks(2, 2048, 16, po.CALC_ISO_GAUSSIAN, dtype=capi.F32, scheme=scheme))
    case("iso6144x24_G3_%s" % _scheme)(lambda scheme=_scheme: run_ranks(3, 6144, 24, po.CALC_ISO_GAUSSIAN, scheme=scheme, runs=((4, 1), (3, 3))))
# whole slices all-gathered (MCMCPP_HIP_COMM_COMPACT=0): what a communicator of more than one rank did before it learned
# to send moved rows only
for _scheme in ("step", "half"):
    case("iso8192x64_G4_%s_whole_slices" % _scheme)(lambda scheme=_scheme: run_ranks(
        4, 8192, 64, po.CALC_ISO_GAUSSIAN, scheme=scheme, env={"MCMCPP_HIP_COMM_COMPACT": "0"},
        expect=lambda st: [] if all(x[2] == 0 and x[1] == 0 and x[0] > 0 for x in st) else ["statistics %r" % (st,)]))


def _expect_repeats(st):
    return [] if all(x[1] >= 1 for x in st) else ["no chunk was repeated although the blocks were too small: %r" % (st,)]


def _expect_learned(full_slots, full_bytes):
    def check(st):
        last = st[-1]
        bad = []
        if not (0 < last[2] < full_slots // 2):
            bad.append("the slot bound was not learned: %r (a whole slice: %d)" % (st, full_slots))
        if not (0 < last[0] < full_bytes / 2):
            bad.append("bytes per step %r against %d for whole slices" % (st, full_bytes))
        return bad
    return check


# blocks that are too small on purpose (8 slots; chunks of 3 steps): every chunk overflows, is rolled back and repeated --
# with stored steps handed out only from chunks that held, an odd number of steps per run, both schemes
for _scheme in ("step", "half"):
    case("forced_overflow_G2_%s" % _scheme)(lambda scheme=_scheme: run_ranks(
        2, 4096, 32, po.CALC_DENSE_GAUSSIAN, scheme=scheme, runs=((5, 1), (3, 3), (1, 1)), env={"MCMCPP_HIP_COMM_COMPACT_CAP": "8", "MCMCPP_HIP_COMM_COMPACT_CHUNK": "3"},
        expect=_expect_repeats))
    case("forced_overflow_G8_%s" % _scheme)(lambda scheme=_scheme: run_ranks(
        8, 8192, 64, po.CALC_ISO_GAUSSIAN, scheme=scheme, runs=((7, 2),), chain_on="rank0", env={"MCMCPP_HIP_COMM_COMPACT_CAP": "8", "MCMCPP_HIP_COMM_COMPACT_CHUNK": "4"},
        expect=_expect_repeats))
    # the bound learned from the run: 16 steps with whole-slice blocks, then what the chunk needed plus an eighth; chunks
    # of 7 steps so that stored steps (every third) fall on both sides of chunk ends
    case("learned_bound_G4_%s" % _scheme)(lambda scheme=_scheme: run_ranks(
        4, 8192, 64, po.CALC_ISO_GAUSSIAN, scheme=scheme, runs=((14, 3), (5, 2)), env={"MCMCPP_HIP_COMM_COMPACT_CHUNK": "7"},
        expect=_expect_learned((2 if scheme == "step" else 1) * 1024, 3 * 2048 * 65 * 8)))
# only rank 0 stores, and more steps than its staging buffer holds (131 072 x 64: 4 stored steps per 256 MiB): the chunks of
# the run end where rank 0's staging is full, and every rank must cut its run the same way although the others store nothing
case("c5_131072x64_G4_rank0_stores_more_than_a_staging_buffer")(lambda: run_ranks(
    4, 131072, 64, po.CALC_ISO_GAUSSIAN, scheme="step", runs=((7, 1), (3, 2)), chain_on="rank0", seed=0, oracle_threads=8))
case("iso32768x64_G2_rank0_stores_half_scheme")(lambda: run_ranks(
    2, 32768, 64, po.CALC_ISO_GAUSSIAN, scheme="half", runs=((20, 1),), chain_on="rank0", env={"MCMCPP_HIP_COMM_COMPACT_CHUNK": "6"}))
# slices that fill neither a wavefront nor a pack workgroup, rows that are not a multiple of 16 bytes
for _scheme in ("step", "half"):
    case("rosen296x3_G2_%s" % _scheme)(lambda scheme=_scheme: run_ranks(2, 296, 3, po.CALC_ROSENBROCK, scheme=scheme, runs=((25, 2), (7, 1))))
    case("iso340x5_f32_G5_%s" % _scheme)(lambda scheme=_scheme: run_ranks(5, 340, 5, po.CALC_ISO_GAUSSIAN, dtype=capi.F32, scheme=scheme, runs=((30, 1),)))
case("one_rank_fails_before_the_first_launch")(lambda: run_ranks(4, 4096, 32, po.CALC_ISO_GAUSSIAN, runs=((2, 2),), bad_rank=2))
# BASELINE config 5 at full size, eight ranks (8 192 walkers of each colour per rank): one exchange per ensemble step, then
# the reference's scheme (one per half-step)
case("c5_131072x64_G8_step")(lambda: run_ranks(8, 131072, 64, po.CALC_ISO_GAUSSIAN, scheme="step", runs=((3, 1),), chain_on="rank0", seed=0,
                                              oracle_threads=8))
case("c5_131072x64_G8_half")(lambda: run_ranks(8, 131072, 64, po.CALC_ISO_GAUSSIAN, scheme="half", runs=((2, 1),), chain_on="rank0", seed=0,
                                              oracle_threads=8))


def main():
    if "loopback" not in os.environ.get("MCMCPP_HIP_RCCL_LIB", ""):
        print(json.dumps({"case": "*", "ok": False, "problems": ["MCMCPP_HIP_RCCL_LIB does not name the loop-back library"]}))
        return 2
    names = sys.argv[1:] or list(CASES)
    bad = 0
    for name in names:
        try:
            problems = CASES[name]()
        except Exception:
            problems = [traceback.format_exc()[-1500:]]
        bad += bool(problems)
        print(json.dumps({"case": name, "ok": not problems, "problems": problems[:8]}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
