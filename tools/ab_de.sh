#!/bin/bash
# differential evolution: parity tests, then the launch time at C2 by knob (update launches alone, planning alone, planning in-stream, batch sizes)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_diffevo.py -x -q -m gpu 2>&1 | tail -5 || exit 1
echo "== default"; python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -2
echo "== iso default"; DE_CALC=iso python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== updates alone"; MCMCPP_HIP_DE_DEBUG=1 DE_CALC=iso python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== updates alone, dense"; MCMCPP_HIP_DE_DEBUG=1 DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== planning alone"; MCMCPP_HIP_DE_DEBUG=2 DE_CALC=iso python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== planning in the launch stream"; MCMCPP_HIP_DE_OVERLAP=0 DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
for b in 8 16; do echo "== batch $b"; MCMCPP_HIP_DE_BATCH=$b DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1; done
for r in 4 16; do echo "== scan run $r"; MCMCPP_HIP_DE_SCAN_RUN=$r DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1; done
