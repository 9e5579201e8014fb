#!/bin/bash
# usage (repo root, GPU box): tools/ab_libs.sh <tag> ...  -- the bench with experiment builds libmcmcpp_hip_<tag>.so (make VARIANT=<tag>)
p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-22s %.3e  %.2f ms/step  %.2f us/launch  acc %.4f'%(sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['acceptance_rate']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for tag in "$@"; do
  lib=$PWD/mcmcpp_amd/libmcmcpp_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/mcmcpp_amd/libmcmcpp_hip.so
  MCMCPP_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-live-counters --steps 6 --warmup 2 --no-cpu-baseline --no-chain > gpurun_out/lib_$tag.json 2> gpurun_out/lib_$tag.err; p gpurun_out/lib_$tag.json $tag
done
