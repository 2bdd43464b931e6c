#!/usr/bin/env python
"""HBM traffic per launch of a kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass).
Units/corrections per MI355X_MICROARCH.md: both counters are in KiB (x1024); on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads, so it is doubled (upper bound for narrow accesses); WRITE_SIZE is used as reported."""
import csv
import json
import sys


def mean_counter(path, kernel_substr, counter, min_ns=0):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= min_ns:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(out, *specs):
    res = {}
    for spec in specs:  # workload:kernel:fetch_csv:write_csv
        wl, kern, fcsv, wcsv = spec.split(":")
        # skip the early-exit launches (a few us) so the average is per REAL evaluation
        f, nf = mean_counter(fcsv, kern, "FETCH_SIZE", 8000)
        w, nw = mean_counter(wcsv, kern, "WRITE_SIZE", 8000)
        res[wl] = {"kernel": kern, "fetch_size_kib_raw": f, "write_size_kib_raw": w, "launches_fetch_pass": nf, "launches_write_pass": nw,
                   "hbm_bytes_per_launch": None if f is None or w is None else int((2 * f + w) * 1024),
                   "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), launches >= 8 us only; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
