#!/usr/bin/env python3
"""Fold two rocprofv3 counter passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into the
record bench.py reads for `roofline.traffic`.

    python tools/pmc_summary.py FETCH_DIR WRITE_DIR OUT.json

Per mi355q kernel: average raw counter value per dispatch (KiB) and the number of dispatches.  The record
carries the sha256 of the kernel sources it was taken on, so a later tree with different kernels does
not present these numbers as its own.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fold(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + directory)
    acc = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter or "mi355q::" not in r["Kernel_Name"]:
                    continue
                name = r["Kernel_Name"].split("(")[0]
                a = acc.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    import bench
    fetch, write = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    kernels = {}
    for name, (n, total) in fetch.items():
        wn, wt = write.get(name, (0, 0.0))
        kernels[name] = {"fetch_size_kb_avg": total / n, "write_size_kb_avg": wt / wn if wn else None, "dispatches": n}
    dominant = max((k for k in kernels if "w4a16_burst_kernel" in k), key=lambda k: kernels[k]["fetch_size_kb_avg"])
    rec = {"kernel_source_sha256": bench.kernel_source_digest(), "kernel_sources": list(bench.KERNEL_SOURCES),
           "dominant_kernel": dominant, "kernels": kernels,
           "note": "raw rocprofv3 counter values (KiB per dispatch); on gfx950 FETCH_SIZE counts a 128-byte request of a "
                   "16-byte-per-lane streaming read as 64 bytes, so HBM bytes read = 2 x FETCH_SIZE"}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    d = kernels[dominant]
    print(dominant, "fetch KiB", d["fetch_size_kb_avg"], "write KiB", d["write_size_kb_avg"], "dispatches", d["dispatches"])


if __name__ == "__main__":
    main()
