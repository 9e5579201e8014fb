#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for k in 1 2; do
  for lib in "" "mcmcpp_amd/libmcmcpp_hip_nochain.so"; do
    MCMCPP_HIP_LIB=$lib python bench.py --no-live-counters --walkers 65536 --dims 32 --calc rosenbrock --no-chain --batch 1000 --interval 100 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=[$lib]', d['value'], d['roofline']['avg_launch_us'])"
  done
done
