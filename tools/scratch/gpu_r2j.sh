#!/bin/bash
out=gpurun_out/r2j; mkdir -p $out
python -m pytest tests/test_sampler_gpu.py -q -m gpu --timeout 900 > $out/pytest_sampler.log 2>&1; echo "pytest sampler rc=$?"; tail -8 $out/pytest_sampler.log
