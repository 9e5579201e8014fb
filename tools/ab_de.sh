#!/bin/bash
# differential evolution: parity tests, then the launch time at C2 by knob
cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python -m pytest tests/test_diffevo.py -x -q -m gpu 2>&1 | tail -15 || exit 1
echo "== default"; python tools/bench_diffevo.py 16384 32 4000 2>&1 | tail -2
echo "== without the matrix-core update"; MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS=-1 python tools/bench_diffevo.py 16384 32 4000 2>&1 | tail -1
echo "== iso"; DE_CALC=iso python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
# (the two timing diagnostics exist in an experiment build only: make -C mcmcpp_amd/csrc VARIANT=detiming EXTRA=-DMCMCPP_DE_TIMING_DIAGNOSTICS)
export MCMCPP_HIP_LIB="$GRAFT_REPO_ROOT/mcmcpp_amd/libmcmcpp_hip_detiming.so"
[ -f "$MCMCPP_HIP_LIB" ] || make -C mcmcpp_amd/csrc -j16 VARIANT=detiming EXTRA=-DMCMCPP_DE_TIMING_DIAGNOSTICS > /dev/null
echo "== updates alone, dense"; MCMCPP_HIP_DE_DEBUG=1 DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
echo "== planning alone"; MCMCPP_HIP_DE_DEBUG=2 DE_CALC=dense python tools/bench_diffevo.py 16384 32 2000 2>&1 | tail -1
unset MCMCPP_HIP_LIB
for w in 65536 131072; do echo "== $w walkers"; python tools/bench_diffevo.py $w 32 400 2>&1 | tail -1; done
