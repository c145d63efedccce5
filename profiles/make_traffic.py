#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/traffic.json (HBM bytes per launch
of each stage's kernel, C2 workload).  Units and the gfx950 correction follow
MI355X_MICROARCH.md section HBM: the counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact.
usage: make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

KERNEL_STAGE = {"k_pack": "pack", "k_hist8": "hist8", "k_reduce8": "reduce8", "k_scatter1p": "scatter1",
                "k_hist2": "hist2", "k_hist16": "hist2", "k_scan2": "scan2", "k_scatter2": "scatter2", "k_localsort2": "localsort",
                "k_intersect": "intersect", "k_intersect3": "intersect", "k_intersect3t": "intersect"}
# (the pipelined kernels last: whichever ran is the one that moves the bytes; k_intersect then only takes oversized items)


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, stage in KERNEL_STAGE.items():
        if k in fetch:
            rd = 2.0 * fetch[k] * 1024.0
            wr = write.get(k, 0.0) * 1024.0
            out[stage] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                          "note": "FETCH_SIZE KiB x2 (gfx950 correction) + WRITE_SIZE KiB"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: round(v["bytes_per_launch"] / 1e9, 3) for k, v in out.items()}))


if __name__ == "__main__":
    main()
