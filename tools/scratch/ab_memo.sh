python -m pytest tests/test_model_gpu.py tests/test_f16s_gpu.py tests/test_sampler_gpu.py tests/test_train_gpu.py -q --timeout 900 2>&1 | tail -3
run() { python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd', '$1', d.get('value'), d['ms_per_step'])"; }
run memo
DIMSUM_FORWARD_MEMO=0 run off
run memo
DIMSUM_FORWARD_MEMO=0 run off
