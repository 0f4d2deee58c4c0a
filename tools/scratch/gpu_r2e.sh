#!/bin/bash
out=gpurun_out/r2e; mkdir -p $out
for pad in 0 64 1024 4160; do
 for shape in "--B 64 --D 1152 --L 1024" "--B 256 --D 1024 --L 256" "--B 16 --D 1152 --L 4096"; do
  for v in 0 4; do
   echo "pad=$pad" >> $out/pad.log
   python tools/bench_scan.py --dmajor $shape --variant $v --iters 30 --pad $pad 2>&1 | grep -v amdgpu >> $out/pad.log
  done
 done
 echo "pad=$pad bwd" >> $out/pad.log
 python tools/bench_scan.py --dmajor --bwd --iters 30 --pad $pad 2>&1 | grep -v amdgpu >> $out/pad.log
done
python - <<'P'
import json
pad=None
for l in open("gpurun_out/r2e/pad.log"):
    l=l.strip()
    if l.startswith("pad="): pad=l; continue
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(pad, d["kernel"], d["fwd_variant"], d["shape"], "median %.3f min %.3f frac %.3f" % (d["ms_median"], d["ms_min"], d["frac_of_8TBps"]))
P
