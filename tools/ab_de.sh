#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for r in 2 4 8 16 32; do for w in 1 2; do echo "scan run $r, runs per lane $w"; DE_CALC=iso MCMCPP_HIP_DE_SCAN_RUN=$r MCMCPP_HIP_DE_FIND_WALKERS=$w python tools/bench_diffevo.py 16384 32 2000 2>&1 | grep "iso calc"; done; done
