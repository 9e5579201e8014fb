#!/bin/bash
# C3 (65536 x 32 Rosenbrock, half-step kernels): what the draw wavefront's place costs (experiments)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_c3_draws.log
: > $out
for k in 0 1 0; do echo "== MCMCPP_HIP_NO_DRAW_WAVE=$k" >> $out; MCMCPP_HIP_NO_DRAW_WAVE=$k timeout -k 5 120 python bench.py --no-live-counters --walkers 65536 --calc rosenbrock --steps 30 --warmup 3 --batch 1000 --interval 1000 --no-chain --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e walker-steps/s, launch %.3f us' % (d['value'], d['roofline'].get('avg_launch_us', 0)))" >> $out 2>&1; done
cat $out
