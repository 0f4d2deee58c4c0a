#!/bin/bash
# one-lane-per-state forward scan vs the 4-lanes-per-channel kernel at the low-parallelism shapes
out=gpurun_out/lanes; mkdir -p $out
python -m pytest tests/test_scan_gpu.py -q -x --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for shape in "16 1152 4096" "64 1152 1024" "32 1152 1024" "8 1152 4096" "256 1024 256"; do
  set -- $shape
  for v in 4 16; do
    echo -n "B=$1 D=$2 L=$3 variant=$v: "
    python tools/bench_scan.py --dmajor --B $1 --D $2 --L $3 --variant $v --iters 30 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_median'], d['ms_min'], round(d['algorithmic_GB']/d['ms_median']/8,3))"
  done
done
python tools/bench_scan.py --dmajor --B 16 --D 1152 --L 4096 --variant 16 --train-fwd --iters 30 | tail -1
python tools/bench_scan.py --dmajor --B 16 --D 1152 --L 4096 --variant 4 --train-fwd --iters 30 | tail -1
