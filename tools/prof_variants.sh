#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in pageable-chain no-chain; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v_$v -- python bench.py --no-live-counters --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --$v > gpurun_out/prof_v_$v.log 2>&1
echo $v; head -2 gpurun_out/prof_v_$v/*/*kernel_stats.csv | tail -1 | awk -F, '{print $(NF-6), $(NF-4), $(NF-2), $(NF-1), $NF}'
done
