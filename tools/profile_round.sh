#!/bin/bash
# usage (repo root, on the GPU box): tools/profile_round.sh <tag>
# the default bench line, the rocprofv3 kernel-trace statistics of the same command (shorter), and the HBM traffic
# counters in two separate --pmc passes (never combined with other trace domains); everything under gpurun_out/
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof_$tag.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_${tag}_$c.log 2>&1
done
cut -c1-600 gpurun_out/bench_$tag.json
head -8 gpurun_out/prof_$tag/*/*kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE stretch_full_step_mfma_kernel gpurun_out/pmc_traffic_$tag.json "C2 16384x32 dense Gaussian, one launch per ensemble step (python bench.py --steps 2 --warmup 1)"
