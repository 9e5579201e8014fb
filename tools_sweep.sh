#!/bin/bash
# usage: tools_sweep.sh <tag> ; runs bench variants, prints compact lines
tag=$1
run() { # name, env..., args
  name=$1; shift
  out=gpurun_out/sweep_${tag}_${name}.log
  env "$@" > /dev/null 2>&1
}
p() { python - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if not l: print(sys.argv[1], "FAILED"); sys.exit(0)
d=json.loads(l[-1]); print("%-28s %.3e w-s/s  %.2f us/launch  frac %.3f  acc %.4f" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["acceptance_rate"]))
PY
}
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
for cfg in "dense_p1:MCMCPP_HIP_PASSES=1:--calc dense" "dense_p2:MCMCPP_HIP_PASSES=2:--calc dense" "dense_p4:MCMCPP_HIP_PASSES=4:--calc dense" \
           "iso_p1:MCMCPP_HIP_PASSES=1:--calc iso" "iso_p2:MCMCPP_HIP_PASSES=2:--calc iso" "iso_p4:MCMCPP_HIP_PASSES=4:--calc iso" "iso_p16:MCMCPP_HIP_PASSES=16:--calc iso" \
           "iso_g4:MCMCPP_HIP_GRAPH_STEPS=4:--calc iso" "iso_g128:MCMCPP_HIP_GRAPH_STEPS=128:--calc iso" \
           "iso_64k:X=1:--calc iso --walkers 65536 --batch 400" "rosen_64k:X=1:--calc rosenbrock --walkers 65536 --batch 400" \
           "iso_1M:X=1:--calc iso --walkers 1048576 --batch 100 --interval 100" "iso_1M_p16:MCMCPP_HIP_PASSES=16:--calc iso --walkers 1048576 --batch 100 --interval 100" "iso_1M_p4:MCMCPP_HIP_PASSES=4:--calc iso --walkers 1048576 --batch 100 --interval 100"; do
  name=${cfg%%:*}; rest=${cfg#*:}; ev=${rest%%:*}; args=${rest#*:}
  env $ev timeout -k 10 120 $B $args > gpurun_out/sweep_${tag}_${name}.log 2>&1
  p $name gpurun_out/sweep_${tag}_${name}.log
done
