#!/bin/bash
# the plain half-step kernel with records two passes ahead and partner rows one pass ahead: parity, then launch times
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_chains.py tests/test_split_loopback.py -q -x -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 120 python tools/time_config.py 65536 32 rosenbrock f64 1000 || exit 1
timeout -k 10 120 python tools/time_config.py 131072 64 iso f64 500 || exit 1
timeout -k 10 120 python tools/time_config.py 131072 64 iso f32 500 || exit 1
for w in 65536 131072 262144 1048576; do timeout -k 10 120 python tools/time_config.py $w 32 iso f64 $((16384000 / w > 500 ? 500 : 16384000 / w)) || exit 1; done
timeout -k 10 120 python tools/time_config.py 1048576 32 rosenbrock f64 16 || exit 1
timeout -k 10 120 python tools/time_config.py 1048576 32 iso f32 16 || exit 1
MCMCPP_HIP_FULL_STEP=0 timeout -k 10 120 python tools/time_config.py 16384 32 iso f64 2000 || exit 1
