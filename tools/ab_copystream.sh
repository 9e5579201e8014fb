p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-22s %.3e  %.2f ms/step  %.2f us/launch'%(sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for cs in 0 1 0 1; do
  MCMCPP_HIP_COPY_STREAM=$cs timeout -k 10 120 python bench.py --no-live-counters --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cs_$cs.json 2> gpurun_out/cs_$cs.err; p gpurun_out/cs_$cs.json copystream$cs
done
for W in 32768; do for fs in 1 0; do MCMCPP_HIP_FULL_STEP=$fs MCMCPP_HIP_FULL_STEP_MAX_WALKERS=100000000 timeout -k 10 120 python bench.py --no-live-counters --steps 3 --warmup 1 --no-cpu-baseline --no-chain --walkers $W --batch 500 --interval 500 > gpurun_out/sz_${fs}_$W.json 2>/dev/null; p gpurun_out/sz_${fs}_$W.json dense_full${fs}_$W; MCMCPP_HIP_FULL_STEP=$fs MCMCPP_HIP_FULL_STEP_MAX_WALKERS=100000000 timeout -k 10 120 python bench.py --no-live-counters --steps 3 --warmup 1 --no-cpu-baseline --no-chain --calc iso --walkers $W --batch 500 --interval 500 > gpurun_out/sz_${fs}_$W.json 2>/dev/null; p gpurun_out/sz_${fs}_$W.json iso_full${fs}_$W; done; done
