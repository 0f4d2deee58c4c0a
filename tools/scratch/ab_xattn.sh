#!/bin/bash
# tools/scratch/ab_xattn.sh "<libs>" "<bench_xattn arg sets separated by ;>"  (GPU box): A/B of attention-kernel variant libraries on one box
libs=${1:-"main"}
IFS=';' read -ra sets <<< "${2:---split3}"
for rep in 1 2; do
for set in "${sets[@]}"; do
  for lib in $libs; do
    if [ "$lib" = main ]; then unset DIMSUM_HIP_LIB; else export DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_$lib.so; fi
    echo -n "$lib | $set | "; python3 tools/bench_xattn.py --iters 30 $set | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_median'],4), round(d['ms_min'],4), round(d['TFLOPs_equivalent'],1))"
  done
done
done
