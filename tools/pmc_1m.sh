cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_1m_$c -- python bench.py --no-live-counters --steps 1 --warmup 1 --no-cpu-baseline --no-chain --calc iso --walkers 1048576 --batch 20 --interval 20 > gpurun_out/pmc_1m_$c.log 2>&1
  python tools/pmc.py gpurun_out/pmc_1m_$c | grep -i "half_step\|IsoGaussian"
done
