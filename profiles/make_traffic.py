#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/traffic.json (HBM bytes per launch
of each stage's kernel, C2 workload).  Units and the gfx950 correction follow
MI355X_MICROARCH.md section HBM: the counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact.
`_step` = the bytes of ALL kernels of one step (the profiled command runs exactly one step: --steps 1 --warmup 0
--no-stage-timers), less the kernels that are not part of a step (the copy-rate probe of bench.py, the placement
probes of a context's first sort); `_meta` names the build (hash of the HIP sources) and the round.
usage: make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [round] [bench arguments]"""
import collections
import csv
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# not part of a step: bench.py's copy-rate probe (k_copy16 and its two 1 GB fills), placement probes, the text reader
NOT_A_STEP = ("k_copy16", "__amd_rocclr_", "k_probe_scatter", "k_tx_")

KERNEL_STAGE = {"k_pack": "pack", "k_hist8": "hist8", "k_reduce8": "reduce8", "k_scatter1p": "scatter1",
                "k_hist2": "hist2", "k_hist16": "hist2", "k_scan2": "scan2", "k_scatter2": "scatter2", "k_localsort2": "localsort",
                "k_intersect": "intersect", "k_intersect3": "intersect", "k_intersect3t": "intersect"}
# (the pipelined kernels last: whichever ran is the one that moves the bytes; k_intersect then only takes oversized items)


def load(path, counter, total=None):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name].append(float(r["Counter_Value"]))
    if total is not None:
        for k, v in acc.items():
            total[k] = (len(v), sum(v))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tf, tw = {}, {}
    fetch = load(sys.argv[1], "FETCH_SIZE", tf)
    write = load(sys.argv[2], "WRITE_SIZE", tw)
    out = {}
    from krisp_amd import build as kb
    out["_meta"] = {"source_sha16": kb.source_sha16(), "round": sys.argv[4] if len(sys.argv) > 4 else None,
                    "measured": time.strftime("%Y-%m-%d %H:%M:%S"),
                    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py " +
                               (sys.argv[5] + " " if len(sys.argv) > 5 else "") + "--steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers"}
    kern = {}
    for k in sorted(set(tf) | set(tw)):
        if k.startswith(NOT_A_STEP):
            continue
        rd = 2.0 * tf.get(k, (0, 0.0))[1] * 1024.0
        wr = tw.get(k, (0, 0.0))[1] * 1024.0
        kern[k] = {"launches": tf.get(k, tw.get(k))[0], "read_bytes": rd, "write_bytes": wr}
    out["_step"] = {"bytes_per_step": sum(v["read_bytes"] + v["write_bytes"] for v in kern.values()), "kernels": kern,
                    "note": "every kernel dispatch of the one profiled step; FETCH_SIZE KiB x2 + WRITE_SIZE KiB"}
    for k, stage in KERNEL_STAGE.items():
        if k in fetch:
            rd = 2.0 * fetch[k] * 1024.0
            wr = write.get(k, 0.0) * 1024.0
            out[stage] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                          "note": "FETCH_SIZE KiB x2 (gfx950 correction) + WRITE_SIZE KiB"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: round(v["bytes_per_launch"] / 1e9, 3) for k, v in out.items() if not k.startswith("_")}),
          "step GB:", round(out["_step"]["bytes_per_step"] / 1e9, 3))


if __name__ == "__main__":
    main()
