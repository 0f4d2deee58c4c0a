python -m pytest tests/test_f16s_train_gpu.py tests/test_train_gpu.py tests/test_gemm_gpu.py tests/test_f16s_gpu.py -q --timeout 900 2>&1 | tail -4
run() { python bench.py --mode $1 $2 --steps 10 --warmup 3 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d.get('value'), d['ms_per_step'])"; }
run train "--batch 64" in-kernel
DIMSUM_ROW_FACTORS_KERNEL=1 run train "--batch 64" launches
run block "" in-kernel
DIMSUM_ROW_FACTORS_KERNEL=1 run block "" launches
