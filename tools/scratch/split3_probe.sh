#!/bin/bash
out=gpurun_out/split3; mkdir -p $out
python -m pytest tests/test_split3_gpu.py tests/test_train_gpu.py -q -x --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
for f in 1 0; do
  echo "DIMSUM_SPLIT3_TRAIN=$f"
  DIMSUM_SPLIT3_TRAIN=$f python bench.py --mode block --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('block', d.get('value'), d.get('ms_per_step'))"
  DIMSUM_SPLIT3_TRAIN=$f python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d.get('value'), d.get('ms_per_step'))"
done
