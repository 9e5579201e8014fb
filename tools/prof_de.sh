#!/bin/bash
# rocprofv3 kernel statistics + trace of the differential-evolution launches at C2 (dense Gaussian), under gpurun_out/prof_de
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DE_CALC=dense
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_de -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/prof_de.log 2>&1
head -12 gpurun_out/prof_de/*/*kernel_stats.csv
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_de/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a window in the middle of the last run: name, start, end relative
mid = len(rows) - 400
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + 90]:
    print("%-40s q%-3s %9.2f %9.2f" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
