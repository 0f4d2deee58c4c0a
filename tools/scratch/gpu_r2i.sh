#!/bin/bash
out=gpurun_out/r2i; mkdir -p $out
for i in 1 2; do
python tools/bench_scan.py --dmajor --bwd --iters 30 --no-out-z 2>&1 | grep -v amdgpu >> $out/scan_bwd.log
for v in nopf noae w3; do
DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_$v.so python tools/bench_scan.py --dmajor --bwd --iters 30 --no-out-z 2>&1 | grep -v amdgpu | sed "s/^/$v /" >> $out/scan_bwd.log
done
done
cut -c1-175 $out/scan_bwd.log
DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_w3.so python -m pytest tests/test_scan_gpu.py -q -m gpu -k bwd --timeout 600 2>&1 | tail -2
