#!/bin/bash
out=gpurun_out/xattn; mkdir -p $out
python -m pytest tests/test_xattn_gpu.py tests/test_split3_gpu.py -q -x --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python3 tools/bench_xattn.py; python3 tools/bench_xattn.py --split3; python3 tools/bench_xattn.py --B 64 --L 1024 --hd 72; python3 tools/bench_xattn.py --B 64 --L 1024 --hd 72 --split3
