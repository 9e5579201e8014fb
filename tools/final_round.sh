#!/bin/bash
# usage (repo root, on the GPU box): tools/final_round.sh <tag> [extras] -- everything the round's measurements come from
# (extras: only the differential-evolution / autocorrelation lines behind the bench and split measurements)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
if [ "$2" != "extras" ]; then
tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
echo "profile done"
python bench.py --no-live-counters --steps 100 --no-cpu-baseline --no-secondary --pinned-chain > gpurun_out/bench_${tag}_pinned.json 2>/dev/null
python bench.py --no-live-counters --mode split > gpurun_out/split_${tag}_full.json 2>/dev/null
python bench.py --no-live-counters --mode split --split-walkers 16384 > gpurun_out/split_${tag}_rank.json 2>/dev/null
MCMCPP_HIP_COMM_FULL_STEP=0 python bench.py --no-live-counters --mode split --split-walkers 16384 > gpurun_out/split_${tag}_rank_half.json 2>/dev/null
echo "split done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_de -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/prof_${tag}_de.log 2>&1
fi
python tools/bench_diffevo.py 16384 32 4000 2>&1 | tail -1 > gpurun_out/de_bench_${tag}.txt
for w in 65536 131072; do python tools/bench_diffevo.py $w 32 400 2>&1 | tail -1 >> gpurun_out/de_bench_${tag}.txt; done
echo "de bench done"
python tools/soak_diffevo.py 20000 > gpurun_out/de_soak_${tag}.txt 2>&1
echo "de soak done"
python tools/bench_autocorr.py 2>&1 | tail -1 > gpurun_out/autocorr_bench_${tag}.txt
echo done
