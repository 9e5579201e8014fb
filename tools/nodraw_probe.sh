#!/bin/bash
# TIMING-ONLY build (-DMCMCPP_EXP_NODRAW=1: the late-draws kernel copies its records instead of making new ones; acceptance drops to
# 0.02, the chain is wrong): what the draws cost the 16-walker matrix-core half-step kernel at four wavefronts per SIMD
cd $GRAFT_REPO_ROOT
for v in "" nodraw; do
  echo "== ${v:-default}"
  if [ -n "$v" ]; then export MCMCPP_HIP_LIB=$GRAFT_REPO_ROOT/mcmcpp_amd/libmcmcpp_hip_$v.so; fi
  timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
  timeout -k 10 120 python tools/time_config.py 131072 32 dense f64 500 || exit 1
  timeout -k 10 120 python tools/time_config.py 262144 32 dense f64 200 || exit 1
  timeout -k 10 120 python tools/time_config.py 1048576 32 dense f64 50 || exit 1
done
