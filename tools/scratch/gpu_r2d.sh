#!/bin/bash
out=gpurun_out/r2d; mkdir -p $out
python -m pytest tests/test_scan_gpu.py -q -m gpu --timeout 900 > $out/pytest_scan.log 2>&1; echo "pytest scan rc=$?"; tail -3 $out/pytest_scan.log
for i in 1 2; do
python tools/bench_scan.py --dmajor --bwd --iters 30 >> $out/scan_bwd.log 2>&1
DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_pfend.so python tools/bench_scan.py --dmajor --bwd --iters 30 >> $out/scan_bwd.log 2>&1
done
grep -v amdgpu.ids $out/scan_bwd.log
for shape in "--B 256 --D 1024 --L 256" "--B 64 --D 1152 --L 1024" "--B 16 --D 1152 --L 4096"; do
  for v in 0 2 4; do
    python tools/bench_scan.py --dmajor $shape --variant $v --iters 30 >> $out/scan_fwd.log 2>&1
  done
done
grep -v amdgpu.ids $out/scan_fwd.log
bash tools/pmc_scan.sh $out/pmc_bwd --dmajor --bwd > $out/pmc_bwd.txt 2>&1; tail -45 $out/pmc_bwd.txt
