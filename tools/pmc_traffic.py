#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into the committed
counter file bench.py reports as roofline.traffic:

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <out.json> "<workload note>"

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: both counters are KiB per dispatch;
on gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads, so it is doubled; WRITE_SIZE is
exact for 16-byte streaming stores.  The file carries the SHA-256 of the library sources it was taken from; bench.py
reports the number only while that digest matches the sources it runs from."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(directory, counter, kernel_sub):
    vals = []
    name = None
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    if not vals:
        raise SystemExit("no %s rows for a kernel matching %r under %s" % (counter, kernel_sub, directory))
    return sum(vals) / len(vals), len(vals), name


def main():
    fetch_dir, write_dir, kernel_sub, out, note = sys.argv[1:6]
    import bench
    fetch, nf, name = mean_counter(fetch_dir, "FETCH_SIZE", kernel_sub)
    write, nw, _ = mean_counter(write_dir, "WRITE_SIZE", kernel_sub)
    rec = {
        "kernel": name.split("(")[0].replace("void mcmcpp::", ""),
        "workload": note,
        "dispatches": [nf, nw],
        "fetch_size_kib_per_launch": fetch,
        "write_size_kib_per_launch": write,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled; WRITE_SIZE exact",
        "source_sha256": bench.source_digest(),
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
