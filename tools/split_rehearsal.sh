#!/bin/bash
# Rehearsals of the split-ensemble measurements on a ONE-GPU box (run through gpurun; results under gpurun_out/):
#   1. bench.py --mode split with 8 loop-back ranks: bytes per step with moved rows only against whole slices
#   2. rocprofv3 --kernel-trace --stats of the same: what the pack / scatter launches add per step
#   3. the driver's multi-GPU command (chains + the bounded split leg) with 2 and 4 gloo ranks sharing the GPU
set -e
export TMPDIR=/tmp
OUT=gpurun_out
python -m pytest tests/test_split_loopback.py -q -k exports > $OUT/r03_shim_build.log 2>&1
python bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 10 > $OUT/r03_split_loopback8_moved_rows.json 2> $OUT/r03_split_loopback8_moved_rows.err
MCMCPP_HIP_COMM_COMPACT=0 python bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 10 > $OUT/r03_split_loopback8_whole_slices.json 2> $OUT/r03_split_loopback8_whole_slices.err
python bench.py --no-live-counters --mode split --loopback-ranks 2 --split-walkers 32768 --steps 10 > $OUT/r03_split_loopback2_32768.json 2>> $OUT/r03_split_loopback8_moved_rows.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_split -o split -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-counters --mode split --loopback-ranks 8 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/r03_split_prof.log 2>&1) || true
find $OUT/prof_split -name "*kernel_stats.csv" -exec cp {} $OUT/r03_split_loopback8_kernel_stats.csv \;
for N in 2 4; do
  MCMCPP_BENCH_BACKEND=gloo MCMCPP_BENCH_SPLIT_LEG=loopback MCMCPP_BENCH_NUMA_BIND=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port 2961$N bench.py --gpus $N --steps 20 --warmup 2 > $OUT/r03_rehearsal_chains_plus_split_gloo_N$N.json 2> $OUT/r03_rehearsal_N$N.err
done
tail -c 600 $OUT/r03_split_loopback8_moved_rows.json; echo; tail -c 300 $OUT/r03_rehearsal_chains_plus_split_gloo_N4.json
