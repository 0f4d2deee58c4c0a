#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_block
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_block -- python3 bench.py --mode block --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg > /dev/null 2>&1
f=$(find /tmp/prof_block -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", tot / 4 / 1e6)
for r in rows[:32]:
    n = r["Name"].replace("dimsum::", "")
    if n.startswith("Cijk"): n = "GEMM " + n[5:22] + " " + n[n.find("MT"):n.find("MT") + 14]
    print(f'{n[:80]:80s} {int(r["Calls"])//4:4d}/step {float(r["AverageNs"])/1e3:9.1f} us {r["Percentage"]:>6s}%')
P
