#!/bin/bash
# usage (repo root, GPU box): tools/ab_sizes.sh -- full-step vs half-step launches across ensemble sizes
p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-26s %.3e w-s/s  %.2f us/launch  (%.0f updates/launch)'%(sys.argv[2],d['value'],d['roofline']['avg_launch_us'],d['roofline']['walker_updates_per_launch']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for calc in dense iso; do
 for W in 32768 49152 65536; do
  batch=$(( 16384000 / W )); [ $batch -gt 2000 ] && batch=2000; [ $batch -lt 100 ] && batch=100
  for fs in 1 0; do
   MCMCPP_HIP_FULL_STEP=$fs MCMCPP_HIP_FULL_STEP_MAX_WALKERS=100000000 timeout -k 10 120 python bench.py --no-live-counters --steps 3 --warmup 1 --no-cpu-baseline --no-chain --calc $calc --walkers $W --batch $batch --interval $batch > gpurun_out/sz_${fs}_${calc}_$W.json 2> gpurun_out/sz_${fs}_${calc}_$W.err; p gpurun_out/sz_${fs}_${calc}_$W.json full${fs}_${calc}_$W
  done
 done
done
