#!/bin/bash
# the workgroup's next draws divided by kind among its wavefronts (16-walker matrix-core half-step kernel): parity, launch times
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_chains.py tests/test_split_loopback.py -q -x -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2000 8 || exit 1
for w in 65536 98304 131072 262144 1048576; do timeout -k 10 120 python tools/time_config.py $w 32 dense f64 $((16384000 / w > 500 ? 500 : 16384000 / w)) || exit 1; done
timeout -k 10 120 python tools/time_config.py 131072 32 dense f32 500 || exit 1
timeout -k 10 120 python tools/time_config.py 1048576 32 dense f32 50 || exit 1
