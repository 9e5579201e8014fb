#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: per-kernel mean duration and mean gap to the previous dispatch."""
import csv, glob, sys, collections
for path in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    prev_end = None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][-60:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[name].append(e - s)
        if prev_end is not None: gap[name].append(s - prev_end)
        prev_end = e
    for k in dur:
        d = sorted(dur[k]); g = sorted(gap[k]) or [0]
        print("%-62s n=%6d dur mean %.0f med %d min %d ns | gap-before med %d mean %.0f ns" % (k, len(d), sum(d)/len(d), d[len(d)//2], d[0], g[len(g)//2], sum(g)/len(g)))
