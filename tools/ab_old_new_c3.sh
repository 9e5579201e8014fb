#!/bin/bash
# same-box A/B of the round-1 tree (ab_old/) against the working tree on the half-step kernels: C3 and the C5 ensemble
cd "$GRAFT_REPO_ROOT"
for k in 1 2; do
  for cfg in "--walkers 65536 --dims 32 --calc rosenbrock" "--walkers 131072 --dims 64 --calc iso"; do
    (cd ab_old && python bench.py --no-live-counters $cfg --no-chain --batch 1000 --interval 100 --steps 30 --warmup 3 --no-cpu-baseline) | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', '$cfg', d['value'], d['roofline']['avg_launch_us'])"
    python bench.py --no-live-counters $cfg --no-chain --batch 1000 --interval 100 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', '$cfg', d['value'], d['roofline']['avg_launch_us'])"
  done
done
