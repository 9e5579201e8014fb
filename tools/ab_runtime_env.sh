#!/bin/bash
# does any knob of the HIP runtime move the launch boundary?  (C2 bench, 40 bench steps each; output under gpurun_out/)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_runtime_env.log
: > $out
run() { echo "== $*" >> $out; timeout -k 5 90 env "$@" python bench.py --no-live-counters --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e walker-steps/s, %.2f ms per step, launch %.2f us' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us', 0)))" >> $out 2>&1 || { echo "failed or timed out" >> $out; return 1; }; }
run A=0 && run AMD_OPT_FLUSH=0 && run AMD_OPT_FLUSH=2 && run AMD_OPT_FLUSH=3 && run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 && run DEBUG_HIP_GRAPH_BATCH_SIZE=1024 && run HIP_FORCE_DEV_KERNARG=0 && run ROC_USE_FGS_KERNARG=0 && run A=1
