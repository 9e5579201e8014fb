#!/bin/bash
# differential evolution's update launch with the jitters' stream positions reached by one jump each (no steps): parity, launch times
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_diffevo.py -q -x -m gpu 2>&1 | tail -2 || exit 1
for w in 16384 65536 131072 262144; do python tools/bench_diffevo.py $w 32 $((w > 100000 ? 400 : 2000)) 2>&1 | tail -1 | cut -c1-420; done
for w in 16384 131072; do DE_CALC=iso python tools/bench_diffevo.py $w 32 $((w > 100000 ? 400 : 2000)) 2>&1 | tail -1 | cut -c1-420; done
