#!/usr/bin/env python3
"""VALU utilisation per kernel from one rocprofv3 pass with
  --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU <1 more SQ> GRBM_GUI_ACTIVE --kernel-trace
usage: python tools/pmc_sq_summary.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [out.json]
Units (/opt/skills/guides/MI355X_MICROARCH.md, rocprofv3 section): SQ_* cycle counters are quad-cycles summed over all
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so effective clock = GRBM_GUI_ACTIVE / 8 / kernel time.
valu_util = SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel quad-cycles at that clock): the share of the chip's vector-ALU
issue slots the kernel filled."""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1].rstrip("/") + "/"
    cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)[0]
    kts = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    # durations: the kernel trace when the pass has one, else the timestamps rocprofv3 >= 1.0 puts on every counter row
    src = kts[0] if kts else cc
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(src))}
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not name.startswith("ty::"):
            continue
        key = (name, r["Grid_Size"], r["Dispatch_Id"])
        per[key][r["Counter_Name"]] = float(r["Counter_Value"])
    groups = collections.defaultdict(list)
    for (name, grid, disp), c in per.items():
        if disp in dur:
            groups[(name, grid)].append((dur[disp], c))
    out = {}
    for (name, grid), lst in sorted(groups.items()):
        n = len(lst)
        du = sum(x[0] for x in lst) / n
        c = {k: sum(x[1].get(k, 0.0) for x in lst) / n for k in lst[0][1]}
        if not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_WAVE_CYCLES"):
            continue
        clk = c["GRBM_GUI_ACTIVE"] / du / 8.0
        avail = du * clk / 4.0 * 1024.0
        out[f"{name} grid={grid}"] = {
            "launches": n, "avg_us": round(du / 1e3, 1), "effective_clock_ghz": round(clk, 2),
            "valu_util": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / avail, 3),
            # VALU instructions a wavefront executes in this launch (SQ_INSTS_VALU counts wave-instructions over the chip)
            "valu_insts_per_wavefront": round(c.get("SQ_INSTS_VALU", 0.0) / max(1.0, int(grid) / 64.0), 1),
            "wave_cycles_active": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3),
            "wave_cycles_issue_stall": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3),
            "wave_cycles_parked": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
