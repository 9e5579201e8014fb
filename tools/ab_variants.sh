#!/bin/bash
# usage (repo root, GPU box): tools/ab_variants.sh  -- full-step vs half-step launches for the bench workloads
p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-22s %.3e  %.2f ms  %.2f us/launch frac %.3f acc %.6f'%(sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac'],d['acceptance_rate']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for fs in 1 0; do
 for calc in dense iso rosenbrock; do
  MCMCPP_HIP_FULL_STEP=$fs timeout -k 10 120 python bench.py --no-live-counters --steps 8 --warmup 2 --no-cpu-baseline --calc $calc > gpurun_out/ab_${fs}_$calc.json 2> gpurun_out/ab_${fs}_$calc.err; p gpurun_out/ab_${fs}_$calc.json full${fs}_${calc}
 done
done
