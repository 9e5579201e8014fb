#!/bin/bash
# usage: tools/ab_libs_1m.sh <tag> ... -- half-step kernels at 1 M x 32 isotropic with experiment builds (see ab_libs.sh)
p() { python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1]));print('%-22s %.3e  %.2f us/launch'%(sys.argv[2],d['value'],d['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[2],'FAILED',e)
" $1 $2; }
for tag in "$@"; do
  lib=$PWD/mcmcpp_amd/libmcmcpp_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/mcmcpp_amd/libmcmcpp_hip.so
  MCMCPP_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-live-counters --steps 4 --warmup 1 --no-cpu-baseline --no-chain --calc iso --walkers 1048576 --batch 100 --interval 100 > gpurun_out/lib1m_$tag.json 2> gpurun_out/lib1m_$tag.err; p gpurun_out/lib1m_$tag.json $tag
done
