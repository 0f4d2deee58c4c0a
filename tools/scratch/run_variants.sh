#!/bin/bash
# usage: run_variants.sh "<bench_scan args>" name1 name2 ...   (GPU box)
args=$1; shift
for v in "$@"; do
  echo -n "$v: "
  DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_$v.so python tools/bench_scan.py $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_median'], d['ms_min'])"
done
