#!/bin/bash
# usage: prof_variants.sh "<bench_scan args>" name1 name2 ...   (GPU box): per-kernel average durations from rocprofv3
args=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  rm -rf /tmp/pv_$v
  DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$v -- python3 tools/bench_scan.py --iters 5 $args > /tmp/pv_$v.log 2>&1
  f=$(find /tmp/pv_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'P'
import csv,sys
v,f=sys.argv[1:3]
rows=list(csv.DictReader(open(f)))
out=[]
for r in rows:
    n=r['Name']
    if "ssm_scan" in n:
        out.append(f"{n.split('(')[0].replace('void dimsum::','')[:48]} avg {float(r['AverageNs'])/1e3:.1f}us min {int(r['MinNs'])/1e3:.1f} x{r['Calls']}")
print(v+': '+' | '.join(out))
P
done
