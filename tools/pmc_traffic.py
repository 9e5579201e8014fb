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


# The secondary configurations of bench.py (its `secondary` list), by the substrings their step kernel's name carries.
SECONDARY = {
    "C3": ["stretch_half_step_kernel<double", "RosenbrockFn"],
    "C5_one_gpu": ["stretch_half_step_kernel<double", "IsoGaussianFn"],
    "C4_one_gpu": ["stretch_half_step_mfma_kernel<double", "DenseGaussianFn"],
    "DE_C2": ["de_update_", "DenseGaussianFn"],
    "C2_f32": ["stretch_full_step_mfma_kernel<float", "DenseGaussianFn"],
    "C5_one_gpu_f32": ["stretch_half_step_kernel<float", "IsoGaussianFn"],
}


def mean_counter(directory, counter, kernel_sub, required=True):
    subs = [kernel_sub] if isinstance(kernel_sub, str) else list(kernel_sub)
    vals = []
    name = None
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and all(x in r["Kernel_Name"] for x in subs):
                vals.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    if not vals:
        if not required:
            return None, 0, None
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
    # the secondary configurations, when the passes ran them too (python bench.py without --no-secondary)
    others = {}
    for key, subs in SECONDARY.items():
        f, n1, nm = mean_counter(fetch_dir, "FETCH_SIZE", subs, required=False)
        w, n2, _ = mean_counter(write_dir, "WRITE_SIZE", subs, required=False)
        if f is None or w is None:
            continue
        others[key] = {"kernel": nm.split("(")[0].replace("void mcmcpp::", ""), "dispatches": [n1, n2],
                       "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
                       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    if others:
        rec["secondary"] = others
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
