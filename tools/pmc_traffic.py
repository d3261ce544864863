#!/usr/bin/env python3
"""HBM traffic of the traversal kernels from rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected in separate runs) -> pmc_traffic.json, which bench.py reads to fill
roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of
16-byte-per-lane reads (the access width of every load in these kernels), so it is
doubled; WRITE_SIZE is taken as is.

usage: pmc_traffic.py <pmc dir> <workload> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys


def counting_variant(name):
    """k_trace_deferred<ANY, COUNT, WIDE>: second argument; the other traversal kernels: <COUNT> or <COUNT, ...>."""
    if "<" not in name:
        return False
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    if "k_trace_deferred" in name:
        return len(args) > 1 and args[1] == "true"
    return args[0] == "true"


def main():
    d, workload, out = sys.argv[1:4]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if counting_variant(name):  # run outside the timed region
                continue
            if "k_extend" in name or "k_trace_deferred<false" in name:
                key = "extend"
            elif "k_shadow" in name or "k_trace_deferred<true" in name:
                key = "shadow"
            else:
                key = None
            if key and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[key][r["Counter_Name"]].add(r["Dispatch_Id"])
    res = {"workload": workload, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled (gfx950, 16-B-per-lane reads)"}
    for key in tot:
        n = max(len(disp[key]["FETCH_SIZE"]), 1)
        fetch = tot[key]["FETCH_SIZE"] * 1024.0 * 2.0
        write = tot[key]["WRITE_SIZE"] * 1024.0
        res[key] = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / max(len(disp[key]["WRITE_SIZE"]), 1)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
