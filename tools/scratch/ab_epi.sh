#!/bin/bash
# tools/scratch/ab_epi.sh (GPU box): the GEMM epilogue variants under dimsum_amd/lib/variants against the shipped library: tile timings and the forward
out=gpurun_out/ab_epi; mkdir -p $out
for v in "" oldepi pacenone pace2 pace6 "" oldepi; do
  if [ -n "$v" ]; then export DIMSUM_HIP_LIB=$PWD/dimsum_amd/lib/variants/libdimsum_hip_$v.so; else unset DIMSUM_HIP_LIB; fi
  echo "== ${v:-shipped}"
  python tools/bench_gemm.py --tiles --rounds 3 --inner 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('  ', d['shape'], {k: v['ms_median'] for k, v in d.items() if isinstance(v, dict)})
"
  python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null > $out/fwd_${v:-shipped}.log; python tools/bench_summary.py $out/fwd_${v:-shipped}.log | head -1
done
