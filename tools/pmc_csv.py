#!/usr/bin/env python3
"""Summarise the counter_collection.csv files of tools/pmc_scan.sh: per kernel (name substring) and counter, the average
value per dispatch. usage: pmc_csv.py <dir> [name-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "dimsum")
    acc = defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if pat not in name:
                continue
            key = name.split("(")[0][-60:]
            acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[key] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"))
    for key, m in meta.items():
        print(f"# {key}: vgpr={m[0]} agpr={m[1]} sgpr={m[2]} lds={m[3]} scratch={m[4]} grid={m[5]}")
    for (key, c), v in sorted(acc.items()):
        print(f"{key:60s} {c:24s} avg={sum(v) / len(v):.4g} n={len(v)}")
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[:1]:
        d = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0][-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in d.items():
            print(f"{k:60s} duration_us avg={sum(v) / len(v) / 1e3:.1f} min={min(v) / 1e3:.1f} n={len(v)}")
            g = acc.get((k, "GRBM_GUI_ACTIVE"))
            if g:       # the counter sums the 8 XCDs; pass 1 carries it and this trace: the clock the kernel ran at under the profiler
                print(f"{k:60s} clock_GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) = {sum(g) / len(g) / 8 / (sum(v) / len(v)):.3f}")
            fs, ws = acc.get((k, "FETCH_SIZE")), acc.get((k, "WRITE_SIZE"))
            if fs and ws:   # KB; gfx950 tallies the 128-B requests of wide reads at 64 B (MI355X_MICROARCH.md, HBM section): x 2
                print(f"{k:60s} hbm_GB per launch (FETCH_SIZE x 2 + WRITE_SIZE) = {(2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024 / 1e9:.4f}")
    # the bench's own line of pass 1 (same process as the counters above): HIP-event median and the box's copy bandwidth, BOTH under the
    # counter-collecting profiler (dispatches serialised and bracketed by counter reads: ~+25 % on either) -- their ratio is what carries over
    for f in sorted(glob.glob(os.path.join(root, "p1.log"))):
        for ln in open(f):
            if ln.startswith("{"):
                print("# pass 1, same process, HIP events (under --pmc: compare the ratio frac_of_box_copy, not the times):", ln.strip())


if __name__ == "__main__":
    main()
