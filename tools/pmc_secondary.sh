#!/bin/bash
# usage (repo root, on the GPU box): tools/pmc_secondary.sh <tag>
# HBM traffic counters of the headline launch AND of the secondary configurations' step kernels (C3, C5's ensemble on one
# GPU, 8 chains in one launch, differential evolution): two separate --pmc passes of the bench command with its secondary
# list (0.2 s of stepping each), never combined with other trace domains.  The raw counter tables stay on the box
# (/tmp); gpurun_out/pmc_traffic_<tag>.json is what profiles/ keeps and bench.py reads.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -- python bench.py --no-live-counters --steps 2 --warmup 1 --no-cpu-baseline --secondary-seconds 0.2 > gpurun_out/pmc_${tag}_$c.log 2>&1
  echo "$c pass done"
done
python tools/pmc_traffic.py /tmp/pmc_${tag}_FETCH_SIZE /tmp/pmc_${tag}_WRITE_SIZE "stretch_full_step_mfma_kernel<double" gpurun_out/pmc_traffic_$tag.json "C2 16384x32 dense Gaussian, one launch per ensemble step (python bench.py --no-live-counters --steps 2 --warmup 1 --secondary-seconds 0.2); secondary: the step kernels of bench.py's secondary configurations in the same passes"
{ python tools/pmc.py /tmp/pmc_${tag}_FETCH_SIZE; python tools/pmc.py /tmp/pmc_${tag}_WRITE_SIZE; } > gpurun_out/pmc_${tag}_summary.txt
