python tools/ubench/write_bw.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_scan_gpu.py tests/test_fullsize_gpu.py -q -m gpu --timeout 900 -x 2>&1 | tail -4
DIMSUM_HIP_LIB=$GRAFT_REPO_ROOT/dimsum_amd/lib/variants/libdimsum_hip_w8.so python -m pytest tests/test_scan_gpu.py tests/test_fullsize_gpu.py -q -m gpu --timeout 900 -x 2>&1 | tail -4
bash tools/scratch/ab_scan.sh "main old w8" "--dmajor --bwd --no-out-z;--dmajor --bwd --no-out-z --B 64 --D 1152 --L 1024" 2>&1 | grep -v amdgpu
