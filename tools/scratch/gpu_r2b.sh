#!/bin/bash
# round-2 GPU pass B: new backward kernel -- scan tests first, timings, then the whole gpu suite
out=gpurun_out/r2b; mkdir -p $out
python -m pytest tests/test_scan_gpu.py -q -m gpu --timeout 900 > $out/pytest_scan.log 2>&1; echo "pytest scan rc=$?"
tail -15 $out/pytest_scan.log
python tools/bench_scan.py --dmajor --bwd --iters 20 > $out/scan_bwd.log 2>&1
python tools/bench_scan.py --dmajor --bwd --iters 20 --B 64 --D 1152 --L 1024 >> $out/scan_bwd.log 2>&1
cat $out/scan_bwd.log
python -m pytest tests -q -m gpu --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python tools/scratch/mall_probe.py > gpurun_out/r2b/mall.log 2>&1; cat gpurun_out/r2b/mall.log
