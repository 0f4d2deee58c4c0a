#!/bin/bash
# round-2 GPU pass A: full gpu test suite, default bench line, forward-variant timings
out=gpurun_out/r2a; mkdir -p $out
python -m pytest tests -q -m gpu -x --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
for shape in "--B 256 --D 1024 --L 256" "--B 64 --D 1152 --L 1024" "--B 16 --D 1152 --L 4096"; do
  for v in 0 2 4; do
    python tools/bench_scan.py --dmajor $shape --variant $v --iters 20 >> $out/scan_variants.log 2>&1
  done
done
python tools/bench_scan.py --dmajor --train-fwd --iters 20 >> $out/scan_variants.log 2>&1
python tools/bench_scan.py --dmajor --bwd --iters 20 >> $out/scan_variants.log 2>&1
cat $out/scan_variants.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"
cat $out/bench.log; tail -3 $out/bench.err
