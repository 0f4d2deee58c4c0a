#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) output: per-kernel average duration and PMC counter values.
usage: pmc_report.py <results.db> [name-substring]"""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "dimsum")
    cur = sqlite3.connect(db).cursor()
    tabs = {r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")}
    if "counters_collection" in tabs:
        q = ("select substr(kernel_name,1,70), counter_name, avg(value), count(*), avg(duration), max(vgpr_count), "
             "max(sgpr_count), max(lds_block_size) from counters_collection where kernel_name like ? "
             "group by kernel_name, counter_name")
        rows = list(cur.execute(q, (f"%{pat}%",)))
        for r in rows:
            print(f"{r[0]:70s} {r[1]:28s} avg={r[2]:.4g} n={r[3]} dur_ns={r[4]:.0f} vgpr={r[5]} sgpr={r[6]} lds={r[7]}")
        if rows:
            return
    if "top_kernels" in tabs:
        for r in cur.execute("select substr(name,1,90), total_calls, total_duration, average, percentage from top_kernels"):
            print(f"{r[0]:90s} calls={r[1]} total_us={r[2]:.1f} avg_us={r[3]:.2f} pct={r[4]:.1f}")


if __name__ == "__main__":
    main()
