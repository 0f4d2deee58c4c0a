#!/usr/bin/env python3
"""tools/kstats.py <kernel_stats.csv> <forwards>: per-kernel average / share / per-forward time of a rocprofv3 --stats summary"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    name = r["Name"]
    short = name[:78] if not name.startswith("Cijk") else name.split("_UserArgs_")[0][-30:] + " " + name.split("_UserArgs_")[1][:28]
    print(f"{short:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / tot * 100:5.1f}% per-fwd {float(r['TotalDurationNs']) / n / 1e6:7.2f} ms")
print("total per forward ms", tot / n / 1e6)
