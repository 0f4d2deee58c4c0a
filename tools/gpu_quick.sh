#!/bin/bash
# tools/gpu_quick.sh (GPU box): the test files named on the command line (default: the f16s / model / attention files), smoke(), the driver bench line
out=gpurun_out/quick; mkdir -p $out
files=${@:-tests/test_f16s_gpu.py tests/test_model_gpu.py tests/test_xattn_gpu.py}
python -m pytest $files -q -m gpu --timeout 900 -s > $out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|vs reference golden|deviation" $out/pytest.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python tools/bench_summary.py $out/bench.log
