#!/bin/bash
# usage (repo root, on the GPU box): tools/profile_round.sh <tag>
# the default bench line, the rocprofv3 kernel-trace statistics of the same command (shorter), and the HBM traffic
# counters in two separate --pmc passes (tools/pmc_secondary.sh; never combined with other trace domains); everything under gpurun_out/
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python bench.py --no-live-counters --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof_$tag.log 2>&1
# kernel durations of the secondary configurations' step kernels (the same command with its secondary list, short)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_secondary -- python bench.py --no-live-counters --steps 2 --warmup 1 --no-cpu-baseline --secondary-seconds 0.5 > gpurun_out/prof_${tag}_secondary.log 2>&1
# the two counter passes (headline launch and the secondary configurations' step kernels) -> gpurun_out/pmc_traffic_$tag.json
tools/pmc_secondary.sh $tag
cut -c1-600 gpurun_out/bench_$tag.json
cp /tmp/prof_$tag/*/*kernel_stats.csv gpurun_out/bench_${tag}_kernel_stats.csv; cp /tmp/prof_${tag}_secondary/*/*kernel_stats.csv gpurun_out/bench_${tag}_secondary_kernel_stats.csv; head -8 gpurun_out/bench_${tag}_kernel_stats.csv | cut -c1-200
