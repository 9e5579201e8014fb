#!/bin/bash
# where the 16-walker matrix-core half-step kernel should switch to late draws (four wavefronts per SIMD): launches of
# 32 768 .. 65 536 updates with the switch forced either way; then the small-launch paths the rework touched
cd $GRAFT_REPO_ROOT
for w in 65536 73728 81920 90112 98304 114688 131072; do
  for late in -1 0; do
    echo "== $w walkers, MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS=$late"
    MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS=$late timeout -k 10 120 python tools/time_config.py $w 32 dense f64 500 || exit 1
  done
done
echo "== C2 by half-steps (8 walkers per wavefront + draw wavefront), C2 by full steps"
MCMCPP_HIP_FULL_STEP=0 timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 || exit 1
timeout -k 10 120 python tools/time_config.py 32768 32 dense f64 1000 || exit 1
echo "== differential evolution (kernel as before)"
python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
