#!/bin/bash
# tools/prof_fwd.sh <name> [bench args]  (GPU box): rocprofv3 kernel-trace + stats of the one-stream headline forward -> gpurun_out/<name>_kernel_stats.csv
name=${1:-fwd}; shift
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DIMSUM_BRANCH_STREAMS=0
rm -rf /tmp/prof_$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 bench.py --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe "$@" > $out/${name}.log 2>&1
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$out/${name}_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    w.writerow([c[:200] for c in r])
P
tail -1 $out/${name}.log | cut -c1-200
