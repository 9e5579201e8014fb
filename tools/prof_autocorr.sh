#!/bin/bash
# autocorrelation kernels: parity tests, the end-to-end bench, rocprofv3 kernel statistics (gpurun_out/prof_ac)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_autocorr.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_autocorr.py 2>&1 | tail -3
rm -rf gpurun_out/prof_ac
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ac -- python tools/bench_autocorr.py > gpurun_out/prof_ac.log 2>&1
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_ac/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)): print(r['Name'][:70], r['Calls'], r['AverageNs'])
"
