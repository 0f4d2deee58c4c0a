python -m pytest tests/test_f16s_train_gpu.py tests/test_train_gpu.py tests/test_f16s_gpu.py tests/test_model_gpu.py -q --timeout 900 2>&1 | tail -3
run() { python bench.py --mode train --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', '$1', d.get('value'), d['ms_per_step'])"; }
run scope
DIMSUM_FORWARD_SCOPE_TRAIN=0 run per-weight
run scope
DIMSUM_FORWARD_SCOPE_TRAIN=0 run per-weight
