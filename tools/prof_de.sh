#!/bin/bash
# rocprofv3 kernel statistics + trace of the differential-evolution launches at C2 (dense Gaussian), under gpurun_out/prof_de
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DE_CALC=${DE_CALC:-dense}
rm -rf gpurun_out/prof_de
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_de -- python tools/bench_diffevo.py 16384 32 500 > gpurun_out/prof_de.log 2>&1
head -12 gpurun_out/prof_de/*/*kernel_stats.csv | cut -c1-250
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_de/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) - 400
t0 = int(rows[mid]["Start_Timestamp"])
prev_end = t0
for r in rows[mid:mid + 72]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s grid %-8s start %9.2f  dur %6.2f  gap %5.2f" % (r["Kernel_Name"][:30], r.get("Grid_Size_X", r.get("Grid_Size", "?")), (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3))
    prev_end = en
PY
