#!/bin/bash
# Rehearsals of the split-ensemble measurements on a ONE-GPU box (run through gpurun; results under gpurun_out/):
#   1. bench.py --mode split with 8 loop-back ranks: bytes per step with moved rows only against whole slices
#   2. rocprofv3 --kernel-trace --stats of the same: what the pack / scatter launches add per step
#   3. the driver's multi-GPU command (chains + the bounded split leg) with 2 and 4 gloo ranks sharing the GPU, the leg through
#      the loop-back library (rank 0 steps all ranks as threads)
#   4. the same command with the leg as the driver will run it -- every rank of it in a child process over real RCCL: on one GPU
#      RCCL refuses two ranks on one device, so this rehearses the FAILURE path: the children fail, the headline line survives
#   5. one child rank of a one-rank real-RCCL communicator (the child's own code path)
set -e
export TMPDIR=/tmp
OUT=gpurun_out
python -m pytest tests/test_split_loopback.py -q -k exports > $OUT/r03_shim_build.log 2>&1
python bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 10 > $OUT/r03_split_loopback8_moved_rows.json 2> $OUT/r03_split_loopback8_moved_rows.err
MCMCPP_HIP_COMM_COMPACT=0 python bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 10 > $OUT/r03_split_loopback8_whole_slices.json 2> $OUT/r03_split_loopback8_whole_slices.err
python bench.py --no-live-counters --mode split --loopback-ranks 2 --split-walkers 32768 --steps 10 > $OUT/r03_split_loopback2_32768.json 2>> $OUT/r03_split_loopback8_moved_rows.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_split -o split -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/r03_split_prof.log 2>&1) || true
find /tmp/prof_split -name "*kernel_stats.csv" -exec cp {} $OUT/r03_split_loopback8_kernel_stats.csv \;
for N in 2 4; do
  MCMCPP_BENCH_BACKEND=gloo MCMCPP_BENCH_SPLIT_LEG=loopback MCMCPP_BENCH_NUMA_BIND=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port 2961$N bench.py --gpus $N --steps 20 --warmup 2 > $OUT/r03_rehearsal_chains_plus_split_gloo_N$N.json 2> $OUT/r03_rehearsal_N$N.err
done
MCMCPP_BENCH_BACKEND=gloo MCMCPP_BENCH_NUMA_BIND=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 20 --warmup 2 > $OUT/r03_rehearsal_children_fail_N2.json 2> $OUT/r03_rehearsal_children_fail_N2.err
# (the id's bootstrap listener lives in the process that made the id: the parent stays alive while its child joins, as in bench.py)
python - > $OUT/r03_split_child_rank.json 2> $OUT/r03_split_child_rank.err <<'PY'
import subprocess, sys
from mcmcpp_amd import capi
cid = capi.comm_unique_id()
r = subprocess.run([sys.executable, "bench.py", "--mode", "split", "--child-rank", "0", "--child-world", "1", "--comm-id-hex", cid.hex(),
                    "--child-device", "0", "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300)
print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "child printed nothing, rc %d: %s" % (r.returncode, r.stderr[-500:]))
PY
python bench.py --no-live-counters --mode split --steps 5 > $OUT/r03_split_one_real_rank.json 2> $OUT/r03_split_one_real_rank.err
tail -c 400 $OUT/r03_rehearsal_children_fail_N2.json; echo; cat $OUT/r03_split_child_rank.json; tail -c 300 $OUT/r03_split_one_real_rank.json
