#!/bin/bash
# tools/scratch/ab_scan.sh "<lib names: main r2 ...>" "<bench_scan arg sets separated by ;>"  (GPU box)
# A/B of kernel-variant libraries on ONE box: every arg set is timed with every library, interleaved (box-to-box spread is 7 %).
libs=${1:-"main"}
IFS=';' read -ra sets <<< "${2:---dmajor --B 64 --D 1152 --L 1024 --variant 4}"
for rep in 1 2; do
for set in "${sets[@]}"; do
  for lib in $libs; do
    if [ "$lib" = main ]; then unset DIMSUM_HIP_LIB; else export DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_$lib.so; fi
    echo -n "$lib | $set | "; python3 tools/bench_scan.py --iters 30 $set | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fwd_variant'], round(d['ms_median'],4), round(d['ms_min'],4), round(d['frac_of_8TBps'],4))"
  done
done
done
