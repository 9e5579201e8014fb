#!/bin/bash
# usage (from the repo root, on the GPU box): tools/sweep.sh <tag> < tools/sweep_walkers.txt
# each input line: <name> <ENV=VALUE> <bench.py arguments...>; prints one compact line per run
tag=$1
p() { python - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")]
if not l: print(sys.argv[1], "FAILED"); sys.exit(0)
d=json.loads(l[-1]); print("%-28s %.3e w-s/s  %.2f ms/step  %.2f us/launch  frac %.3f  acc %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["acceptance_rate"]))
PY
}
B="python bench.py --no-live-counters --steps 4 --warmup 1 --no-cpu-baseline"
while read -r name ev args; do
  [ -z "$name" ] && continue
  env $ev timeout -k 10 120 $B $args > gpurun_out/sweep_${tag}_${name}.log 2>&1
  p $name gpurun_out/sweep_${tag}_${name}.log
done
