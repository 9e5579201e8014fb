#!/bin/bash
# same-box A/B of the round-1 tree (ab_old/, built from git) against the working tree: the C2 bench, alternating
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
  (cd ab_old && python bench.py --no-live-counters --steps 40 --warmup 3 --no-cpu-baseline) | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  python bench.py --no-live-counters --steps 40 --warmup 3 --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new pageable', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  python bench.py --no-live-counters --steps 40 --warmup 3 --no-cpu-baseline --no-secondary --pinned-chain | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new pinned', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
