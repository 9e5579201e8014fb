#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for w in 1 2; do echo "find batches per wave $w"; MCMCPP_HIP_DE_FIND_WALKERS=$w python tools/bench_diffevo.py 16384 32 2000 2>&1 | grep -o "device.*roofline[^;]*" | cut -c1-220; done
for d in 1 2; do echo "debug $d"; MCMCPP_HIP_DE_DEBUG=$d python tools/bench_diffevo.py 16384 32 2000 2>&1 | grep -o "device.*roofline[^;]*" | cut -c1-220; done
