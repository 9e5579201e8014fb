#!/bin/bash
# usage (repo root, GPU box): tools/sweep_r03.sh  -- the stretch path from 4 096 to 1 048 576 walkers x 32 dims, three targets, both
# element types, with the library's own kernel choice (tools/time_config.py: us per step launch by HIP events, nominal roofline fraction)
cd $GRAFT_REPO_ROOT
export TIME_CONFIG_SECONDS=0.4
for calc in dense iso rosenbrock; do
  for dt in f64 f32; do
    for w in 4096 16384 32768 65536 131072 262144 1048576; do
      b=$(( 16384000 / w )); [ $b -gt 2000 ] && b=2000
      python tools/time_config.py $w 32 $calc $dt $b 2>&1 | grep -v amdgpu.ids | cut -c1-170
    done
  done
done
