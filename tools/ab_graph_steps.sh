p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-22s %.3e  %.3f ms/step  %.2f us/launch'%(sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for gs in 128 100 128 100; do
  MCMCPP_HIP_GRAPH_STEPS=$gs timeout -k 10 120 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/gs_$gs.json 2> gpurun_out/gs_$gs.err; p gpurun_out/gs_$gs.json graph_steps_$gs
done
