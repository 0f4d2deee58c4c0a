#!/bin/bash
# kernel-trace + stats of the headline forward (one stream) -> gpurun_out/prof_f16s/
out=$GRAFT_REPO_ROOT/gpurun_out/prof_f16s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DIMSUM_BRANCH_STREAMS=0
rm -rf /tmp/prof_f
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python3 bench.py --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe > $out/bench.log 2>&1
f=$(find /tmp/prof_f -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$out/kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    w.writerow([c[:200] for c in r])
P
tail -1 $out/bench.log | cut -c1-200
