#!/bin/bash
# the headline kernel (stretch_full_step_mfma_kernel) addressed by (uniform base, 32-bit byte offset): parity, launch times
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_chains.py tests/test_split_loopback.py tests/test_async_pinned.py -q -x -m gpu 2>&1 | tail -3 || exit 1
for i in 1 2; do
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2400 || exit 1
done
timeout -k 10 120 python tools/time_config.py 16384 32 dense f32 2400 || exit 1
timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2400 2 || exit 1
timeout -k 10 120 python tools/time_config.py 8192 32 dense f64 2400 || exit 1
timeout -k 10 120 python tools/time_config.py 4096 32 dense f64 2400 || exit 1
