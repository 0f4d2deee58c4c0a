#!/bin/bash
# tools/scratch/prof_fwd.sh <name> <bench args...>: kernel-trace + stats of one bench invocation (GPU box), summary to gpurun_out/prof/<name>_kernel_stats.csv
name=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf /tmp/prof_$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 bench.py "$@" > $R/gpurun_out/prof/$name.log 2>&1
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" "$R/gpurun_out/prof/${name}_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    w.writerow([c[:200] for c in r])
P
tail -1 $R/gpurun_out/prof/$name.log | cut -c1-200
