#!/bin/bash
# launch times of the dense (matrix-core) step kernels at the sizes where they matter (tools/time_config.py)
cd $GRAFT_REPO_ROOT
python tools/time_config.py 16384 32 dense f64 2000 8
python tools/time_config.py 65536 32 dense f64 500
python tools/time_config.py 131072 32 dense f64 500
python tools/time_config.py 262144 32 dense f64 200
python tools/time_config.py 131072 32 dense f32 500
python tools/time_config.py 32768 32 dense f64 1000
MCMCPP_HIP_FULL_STEP=0 python tools/time_config.py 16384 32 dense f64 2000
python tools/time_config.py 16384 32 dense f64 2000
