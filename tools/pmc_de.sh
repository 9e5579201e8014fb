#!/bin/bash
# differential evolution at C2 (dense Gaussian): where the wavefront cycles go, and the HBM traffic of the two kernels
# (separate rocprofv3 --pmc passes, never combined with other trace domains); summaries under gpurun_out/pmc_de_*.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DE_CALC=dense
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_de_sq -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/pmc_de_sq.log 2>&1
python tools/pmc.py gpurun_out/pmc_de_sq | grep "de_" > gpurun_out/pmc_de_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_de_$c -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/pmc_de_$c.log 2>&1
  python tools/pmc.py gpurun_out/pmc_de_$c | grep "de_" >> gpurun_out/pmc_de_summary.txt
done
cat gpurun_out/pmc_de_summary.txt
