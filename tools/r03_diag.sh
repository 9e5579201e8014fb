#!/bin/bash
# usage (repo root, GPU box): tools/r03_diag.sh <tag>  -- kernel durations (--kernel-trace --stats) and the five SQ counters
# (a pass of its own, never combined with other trace domains) of the bench command WITH its secondary list.
# Traces go to /tmp (they are large); the per-kernel statistics and the counter summary to gpurun_out/.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/diag_${tag}_stats -- python bench.py --no-live-counters --steps 2 --warmup 1 --no-cpu-baseline --secondary-seconds 0.5 > gpurun_out/diag_${tag}_stats.log 2>&1
cp /tmp/diag_${tag}_stats/*/*kernel_stats.csv gpurun_out/diag_${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/diag_${tag}_sq -- python bench.py --no-live-counters --steps 2 --warmup 1 --no-cpu-baseline --secondary-seconds 0.2 > gpurun_out/diag_${tag}_sq.log 2>&1
python tools/pmc.py /tmp/diag_${tag}_sq > gpurun_out/diag_${tag}_sq.txt
cut -c1-120 gpurun_out/diag_${tag}_kernel_stats.csv | head -12
