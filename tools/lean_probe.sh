#!/bin/bash
# TIMING-ONLY experiment builds of the dense matrix-core half-step kernel (results are wrong where the next draws are skipped):
#   default        192 VGPRs, P^T in registers, next draws in the gather's shadow
#   leand          no next draws                       (172 VGPRs)
#   leanb          no next draws, P^T through LDS      (160 VGPRs)
#   leanc          the same held to 128 VGPRs by the compiler (84-92 bytes of spills): four wavefronts per SIMD
cd $GRAFT_REPO_ROOT
for v in "" leand leanb leanc; do
  echo "== variant ${v:-default}"
  if [ -n "$v" ]; then export MCMCPP_HIP_LIB=$GRAFT_REPO_ROOT/mcmcpp_amd/libmcmcpp_hip_$v.so; fi
  timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
  timeout -k 10 120 python tools/time_config.py 131072 32 dense f64 500 || exit 1
  timeout -k 10 120 python tools/time_config.py 262144 32 dense f64 200 || exit 1
done
