#!/bin/bash
# draw records made ahead in batches (MCMCPP_HIP_BATCH_DRAWS) against the launches making their own: the C2 launch time
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_libs2.log
: > $out
for b in 128 0 32 256 300 150 128; do echo "== MCMCPP_HIP_BATCH_DRAWS=$b" >> $out; MCMCPP_HIP_BATCH_DRAWS=$b timeout -k 5 90 python bench.py --no-live-counters --steps 60 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e walker-steps/s, %.2f ms per step, launch %.3f us, acceptance %.4f' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us', 0), d['acceptance_rate']))" >> $out 2>&1; done
cat $out
