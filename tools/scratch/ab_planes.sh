for i in 1 2; do
for v in 1 0 auto; do
DIMSUM_OUT_PROJ_PLANES=$v timeout 300 python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('planes=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done; done
DIMSUM_OUT_PROJ_PLANES=1 timeout 300 python bench.py --mode xl512 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl planes=1', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
DIMSUM_OUT_PROJ_PLANES=0 timeout 300 python bench.py --mode xl512 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xl planes=0', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
