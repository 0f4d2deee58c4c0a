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


if __name__ == "__main__":
    main()
