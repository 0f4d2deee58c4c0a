#!/bin/bash
out=gpurun_out/split3; mkdir -p $out
python -m pytest tests/test_split3_gpu.py tests/test_model_gpu.py tests/test_sampler_gpu.py -q -x --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
DIMSUM_SPLIT3_MIN_ROWS=0 python -m pytest tests/test_model_gpu.py tests/test_token_ops_gpu.py tests/test_xattn_gpu.py -q -x --timeout 900 > $out/pytest2.log 2>&1; echo "pytest(min rows 0) rc=$?"; tail -3 $out/pytest2.log
for f in 1 0; do
  echo "DIMSUM_SPLIT3=$f"; DIMSUM_SPLIT3=$f python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  DIMSUM_SPLIT3=$f python bench.py --mode xl512 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl512', d.get('value'), d.get('ms_per_step'))"
done
