p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-30s %.3e w-s/s  %.2f us/launch'%(sys.argv[2],d['value'],d['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for calc in rosenbrock iso dense; do for W in 32768 65536 131072; do for fs in 1 0; do
  MCMCPP_HIP_FULL_STEP=$fs MCMCPP_HIP_FULL_STEP_MAX_WALKERS=100000000 timeout -k 10 120 python bench.py --no-live-counters --steps 6 --warmup 2 --no-cpu-baseline --no-chain --calc $calc --walkers $W --batch 250 --interval 250 > gpurun_out/c3_$fs.json 2>/dev/null; p gpurun_out/c3_$fs.json ${calc}_${W}_full$fs
done; done; done
