#!/bin/bash
out=gpurun_out/r2g; mkdir -p $out
V=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_ldsr.so
DIMSUM_HIP_LIB=$V python -m pytest tests/test_scan_gpu.py tests/test_fullsize_gpu.py -q -m gpu --timeout 900 -k "bwd or scan_fwd_bwd" > $out/pytest_scan.log 2>&1; echo "pytest scan rc=$?"; tail -3 $out/pytest_scan.log
for i in 1 2 3; do
python tools/bench_scan.py --dmajor --bwd --iters 30 --no-out-z >> $out/scan_bwd.log 2>&1
DIMSUM_HIP_LIB=$V python tools/bench_scan.py --dmajor --bwd --iters 30 --no-out-z >> $out/scan_bwd.log 2>&1
done
python tools/bench_scan.py --dmajor --bwd --iters 30 >> $out/scan_bwd.log 2>&1
DIMSUM_HIP_LIB=$V python tools/bench_scan.py --dmajor --bwd --iters 30 >> $out/scan_bwd.log 2>&1
python tools/bench_scan.py --dmajor --bwd --iters 30 --B 64 --D 1152 --L 1024 >> $out/scan_bwd.log 2>&1
DIMSUM_HIP_LIB=$V python tools/bench_scan.py --dmajor --bwd --iters 30 --B 64 --D 1152 --L 1024 >> $out/scan_bwd.log 2>&1
grep -v amdgpu.ids $out/scan_bwd.log | cut -c1-200
