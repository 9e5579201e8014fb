#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DE_CALC=iso
for d in 1 2; do
export MCMCPP_HIP_DE_DEBUG=$d
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_de_$d -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/pmc_de_$d.log 2>&1
echo "debug $d"; python tools/pmc.py gpurun_out/pmc_de_$d | grep de_
done
