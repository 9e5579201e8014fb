#!/bin/bash
# one launch per ENSEMBLE step (full-step kernels) beyond 32 768 walkers per launch: C4 on one GPU and large single ensembles
cd $GRAFT_REPO_ROOT
export MCMCPP_HIP_FULL_STEP_MAX_WALKERS=100000000
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 4 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 2 || exit 1
timeout -k 10 120 python tools/time_config.py 65536 32 dense f64 500 || exit 1
timeout -k 10 120 python tools/time_config.py 131072 32 dense f64 500 || exit 1
timeout -k 10 120 python tools/time_config.py 65536 32 rosenbrock f64 500 || exit 1
timeout -k 10 120 python tools/time_config.py 131072 64 iso f64 500 || exit 1
unset MCMCPP_HIP_FULL_STEP_MAX_WALKERS
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 4 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 2 || exit 1
