#!/bin/bash
# differential evolution: parity tests, then the launch time at C2 by knob
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python -m pytest tests/test_diffevo.py -x -q -m gpu 2>&1 | tail -15 || exit 1
echo "== default"; python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -2
echo "== iso"; DE_CALC=iso python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== updates alone, dense"; MCMCPP_HIP_DE_DEBUG=1 DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
for d in 2 3 4 5; do echo "== planning alone, debug $d"; MCMCPP_HIP_DE_DEBUG=$d DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1; done
for r in 4 16; do echo "== scan run $r"; MCMCPP_HIP_DE_SCAN_RUN=$r DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1; done
for b in 16 32; do echo "== batch $b"; MCMCPP_HIP_DE_BATCH=$b DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1; done
