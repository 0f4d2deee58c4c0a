#!/bin/bash
# usage: prof_fwd.sh "<bench_scan args>" : rocprofv3 kernel durations of the forward scan under each kernel variant (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 0 2 4; do
  rm -rf /tmp/pf_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$s -- python3 tools/bench_scan.py --iters 10 --variant $s $1 > /tmp/pf_$s.log 2>&1
  f=$(find /tmp/pf_$s -name "*kernel_stats.csv" | head -1)
  python3 - "$s" "$f" <<'P'
import csv,sys
v,f=sys.argv[1:3]
for r in csv.DictReader(open(f)):
    if 'ssm_scan' in r['Name']:
        print(f"variant={v} {r['Name'].split('(')[0][-70:]} avg {float(r['AverageNs'])/1e3:.1f}us min {int(r['MinNs'])/1e3:.1f} x{r['Calls']}")
P
done
