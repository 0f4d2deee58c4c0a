run() { python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd', '$1', d.get('value'), d['ms_per_step'])"; }
run fix
DIMSUM_TT_FIX=0 run runtime-checks
run fix
DIMSUM_TT_FIX=0 run runtime-checks
python bench.py --mode xl512 --steps 10 --warmup 3 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl512 fix', d.get('value'), d['ms_per_step'])"
DIMSUM_TT_FIX=0 python bench.py --mode xl512 --steps 10 --warmup 3 --no-cpu-baseline --no-box-probe --no-fp32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl512 runtime-checks', d.get('value'), d['ms_per_step'])"
