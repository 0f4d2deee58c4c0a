python -m pytest tests/test_scan_gpu.py tests/test_fullsize_gpu.py tests/test_train_gpu.py -q -m gpu --timeout 900 2>&1 | tail -2
for i in 1 2; do
python tools/bench_scan.py --dmajor --bwd --iters 30 --no-out-z 2>&1 | grep -v amdgpu | cut -c90-160
python tools/bench_scan.py --dmajor --bwd --iters 30 2>&1 | grep -v amdgpu | cut -c90-160 | sed 's/^/outz /'
done
