#!/bin/bash
# 8 against 16 walkers per wavefront (matrix-core half-step kernel) below the present switch at 32 768 updates per launch
cd $GRAFT_REPO_ROOT
for w in 32768 40960 49152 57344 65536; do
  for four in 1000000 1; do
    echo "== $w walkers, MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=$four"
    MCMCPP_HIP_FULL_STEP=0 MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=$four timeout -k 10 120 python tools/time_config.py $w 32 dense f64 500 || exit 1
  done
done
