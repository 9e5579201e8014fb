#!/bin/bash
# the draw records of the next graph replay made on a parallel branch of this one (MCMCPP_HIP_FILL_BRANCH = pieces): parity, then
# the headline launch time against the number of pieces (0: one fill launch in line at the head of every replay)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "branch_of_this_one or batches_of_any_length" 2>&1 | tail -3 || exit 1
for p in 0 1 4 10 30 0 10; do
  echo "== MCMCPP_HIP_FILL_BRANCH=$p"
  MCMCPP_HIP_FILL_BRANCH=$p timeout -k 10 120 python tools/time_config.py 16384 32 dense f64 2400 || exit 1
done
