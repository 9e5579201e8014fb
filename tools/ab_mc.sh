#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
  for mc in 0 1; do
    MCMCPP_HIP_FORCE_MC=$mc python bench.py --no-live-counters --steps 40 --warmup 3 --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force_mc=$mc', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
