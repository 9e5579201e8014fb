# usage (GPU box): bash tools/ab_graph_steps.sh   -- bench line for several graph lengths / ring sizes, interleaved
p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-28s %.3e  %.3f ms/step  %.2f us/launch'%(sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for rep in 1 2; do
for cfg in "128 32" "300 32" "400 64" "800 128"; do
  set -- $cfg
  MCMCPP_HIP_GRAPH_STEPS=$1 MCMCPP_HIP_CHAIN_SUBCHUNK_MB=$2 timeout -k 10 120 python bench.py --no-live-counters --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/gs_$1.json 2> gpurun_out/gs_$1.err; p gpurun_out/gs_$1.json graph_${1}_ring_mb_$2
done
done
