#!/bin/bash
# fp32 against fp64 launch times of the stretch kernels (tools/time_config.py), half-step and full-step paths
cd $GRAFT_REPO_ROOT
export TIME_CONFIG_SECONDS=0.5
for dt in f32 f64; do
 MCMCPP_HIP_FULL_STEP=0 python tools/time_config.py 16384 32 iso $dt 500
 python tools/time_config.py 131072 32 iso $dt 500
 python tools/time_config.py 131072 64 iso $dt 500
 python tools/time_config.py 65536 32 rosenbrock $dt 500
 python tools/time_config.py 1048576 32 iso $dt 50
 python tools/time_config.py 16384 32 dense $dt 2000
 python tools/time_config.py 16384 32 iso $dt 2000
done
