#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection csv): mean counter value per dispatch of each kernel."""
import csv, glob, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][-48:]
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in vals.items():
    for c, v in d.items():
        print("%-50s %-12s n=%6d mean %.3f  (sum %.1f)" % (k, c, len(v), sum(v) / len(v), sum(v)))
