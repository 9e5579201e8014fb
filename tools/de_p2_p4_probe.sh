#!/bin/bash
# differential evolution's update kernel with 8 (default below 32 768 updates per launch) against 16 walkers per wavefront
cd $GRAFT_REPO_ROOT
for w in 32768 40960 49152 57344; do
  for four in 1000000 1; do
    echo "== $w walkers, MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=$four"
    MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=$four python tools/bench_diffevo.py $w 32 1000 2>&1 | tail -1 | cut -c1-330
  done
done
