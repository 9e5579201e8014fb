#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
  for lib in "" "mcmcpp_amd/libmcmcpp_hip_prio3.so"; do
    MCMCPP_HIP_LIB=$lib python bench.py --no-live-counters --steps 40 --warmup 3 --no-cpu-baseline --no-secondary --no-chain | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=[$lib]', d['value'], d['roofline']['avg_launch_us'])"
  done
done
