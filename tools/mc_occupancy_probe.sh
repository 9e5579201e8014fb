#!/bin/bash
# the matrix-core half-step / DE update kernels after the round-3 rework (32-bit offsets, P^T through LDS, counter adds, stored
# steps re-read): parity, then launch times by walkers per wavefront and by where the next draws / the jitters are made
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_chains.py tests/test_diffevo.py -q -x -m gpu -k "matrix_core or chains or config4 or wave_mapping or diffevo or dense or differential" 2>&1 | tail -3 || exit 1
echo "== stretch, default selection (late draws from 49 153 updates per launch on)"
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
for w in 65536 98304 131072 262144 1048576; do timeout -k 10 120 python tools/time_config.py $w 32 dense f64 $((16384000 / w > 500 ? 500 : 16384000 / w)) || exit 1; done
timeout -k 10 120 python tools/time_config.py 131072 32 dense f32 500 || exit 1
timeout -k 10 120 python tools/time_config.py 1048576 32 dense f32 50 || exit 1
echo "== stretch, late draws never (MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS=-1)"
export MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS=-1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
for w in 98304 131072 1048576; do timeout -k 10 120 python tools/time_config.py $w 32 dense f64 $((16384000 / w > 500 ? 500 : 16384000 / w)) || exit 1; done
unset MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS
echo "== stretch, 8 walkers per wavefront up to 65 536 updates (MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=65537)"
for w in 65536 98304; do MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=65537 timeout -k 10 120 python tools/time_config.py $w 32 dense f64 500 || exit 1; done
echo "== differential evolution, default (16 walkers per wavefront from 32 768 updates on)"
for w in 16384 65536 98304 131072 262144; do python tools/bench_diffevo.py $w 32 $((w > 100000 ? 400 : 2000)) 2>&1 | tail -1; done
echo "== differential evolution, 8 walkers per wavefront throughout"
for w in 65536 98304 131072; do MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS=100000000 python tools/bench_diffevo.py $w 32 400 2>&1 | tail -1; done
