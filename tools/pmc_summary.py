#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (CSV output) per kernel.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
Correction (per /opt/skills/guides/MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled; WRITE_SIZE is taken as is.  Both facts
are re-checked against kernels of known byte counts in the same run (torch fill / bitwise_and of a
32 MiB tensor: WRITE 32768 KiB exact, FETCH 16.4 MB = 1/2)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            # (templated kernels are listed as "void ty::name<20u>(...)")
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    f, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    w, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("ty::"):
            continue
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        out[k] = {"launches": nf.get(k, 0), "fetch_size_kib_raw": fk, "write_size_kib": wk,
                  "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
